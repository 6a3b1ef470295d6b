/* mpecdsa_hip.h — C-ABI of the MI355X (gfx950) batched crypto core for GG20 threshold signing.
 *
 * Drop-in boundary: the reference (ZenGo-X/multi-party-ecdsa v0.8.1) has no FFI of
 * its own on this path; its arithmetic leaves the crate through curv-kzen / kzen-paillier method
 * calls.  Every entry point below is the batched equivalent of one such call and cites the call
 * site(s) it replaces.  A Rust shim crate that re-exports the curv/paillier names would bind these
 * with `extern "C"`.
 *
 * Conventions
 *  - Big integers cross the boundary as fixed-width little-endian arrays of uint32_t words
 *    ("interface words"): 64 words = 2048 bit, 128 words = 4096 bit.  A batch is item-major:
 *    item i occupies words [i*K32, (i+1)*K32).
 *  - All data pointers are DEVICE pointers (HBM-resident); the library never allocates or frees
 *    caller memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *    are asynchronous with respect to the host unless stated otherwise.
 *  - Return value: 0 = ok, negative = MPE_E_* (argument/launch errors).  Verifiers additionally
 *    write a per-item uint8_t ok[] (1 = accept) — the batched form of the reference's
 *    bool / Result<(), Error> returns; one bad item never aborts the batch.
 *  - No global state: distinct mpe_ctx may be used from distinct host threads / streams.
 */
#ifndef MPECDSA_HIP_H
#define MPECDSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPE_OK 0
#define MPE_E_ARG (-1)     /* bad argument (NULL pointer, unsupported width, ...) */
#define MPE_E_HIP (-2)     /* HIP runtime error; see mpe_last_error() */
#define MPE_E_NOMEM (-3)

typedef struct mpe_ctx mpe_ctx;         /* per-thread/stream context: owns scratch (window tables) */
typedef struct mpe_modset mpe_modset;   /* a set of moduli with their Montgomery constants in HBM */

/* Library / device info.  mpe_version() never touches the GPU. */
const char* mpe_version(void);
const char* mpe_last_error(void);

int mpe_ctx_create(mpe_ctx** out, int device);
int mpe_ctx_destroy(mpe_ctx* ctx);
/* Blocks the host until everything queued on `stream` has finished (hipStreamSynchronize). */
int mpe_sync(mpe_ctx* ctx, void* stream);

/* ---- moduli ------------------------------------------------------------------------------ */
/* Precomputes, ON THE GPU, the Montgomery constants of `count` odd moduli of `bits` (2048|4096)
 * bits: n in the kernel's internal radix, -n^-1, R mod n, R^2 mod n.  In the reference these are
 * rebuilt inside every mpz_powm call; per-key reuse is output-identical.
 * d_moduli: [count][bits/32] interface words.  Moduli must be odd and >= 3. */
int mpe_modset_create(mpe_ctx* ctx, int bits, int count, const uint32_t* d_moduli,
                      mpe_modset** out, void* stream);
int mpe_modset_destroy(mpe_modset* ms);
int mpe_modset_count(const mpe_modset* ms);
int mpe_modset_bits(const mpe_modset* ms);

/* ---- BigInt::mod_pow --------------------------------------------------------------------- */
/* out[i] = base[i] ^ exp[i] mod modulus[idx(i)]           (curv `BigInt::mod_pow` == mpz_powm;
 * reference call sites: src/utilities/mta/range_proofs.rs:52-57,86,122-141,238-249,291,339-372;
 * src/utilities/zk_pdl_with_slack/mod.rs:189-195).
 *  d_mod_idx : per-item index into `ms`, or NULL meaning idx(i) = (count==1 ? 0 : i).
 *  d_base    : [batch][bits/32], any value < 2^bits (need not be reduced).
 *  d_exp     : [batch][exp_words] little-endian words, exponent >= 0.
 *  d_out     : [batch][bits/32], canonical residue in [0, n).
 * Fixed 4-bit windows, constant sequence of operations for a given exp_words (exponents on this
 * path are secret nonces). */
int mpe_modexp(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx,
               const uint32_t* d_base, const uint32_t* d_exp, int exp_words, uint32_t* d_out,
               void* stream);

/* out[i] = a[i] * b[i] mod modulus[idx(i)]   (curv `BigInt::mod_mul`,
 * src/utilities/zk_pdl_with_slack/mod.rs:198; also Paillier::add = mulmod N^2, mta/mod.rs:145) */
int mpe_modmul(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx,
               const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, void* stream);

/* ---- kzen-paillier ------------------------------------------------------------------------ */
/* A set of Paillier-2048 keys resident in HBM.  Public sets hold N, N^2 and their Montgomery
 * constants; private sets are built from (p, q) and additionally hold p^2, q^2, h_p, h_q and the CRT
 * constants — all computed ON THE GPU at creation (the reference recomputes them in every
 * Paillier::decrypt call; reuse is output-identical).
 *  d_N: [nkeys][64]   d_p, d_q: [nkeys][32]  (1024-bit primes)
 * key_idx arguments: per-item key index, or NULL meaning (nkeys==1 ? 0 : i). */
typedef struct mpe_paillier mpe_paillier;
int mpe_paillier_create_public(mpe_ctx* ctx, int nkeys, const uint32_t* d_N, mpe_paillier** out, void* stream);
int mpe_paillier_create_private(mpe_ctx* ctx, int nkeys, const uint32_t* d_p, const uint32_t* d_q,
                                mpe_paillier** out, void* stream);
int mpe_paillier_destroy(mpe_paillier* pk);
int mpe_paillier_nkeys(const mpe_paillier* pk);
/* device pointer to the [nkeys][64] table of moduli N (valid for the life of the key set) */
const uint32_t* mpe_paillier_n(const mpe_paillier* pk);

/* c = (1 + m*N) * r^N mod N^2     `Paillier::encrypt_with_chosen_randomness(ek, m, r)`
 * (src/utilities/mta/mod.rs:68-75,133-137).  m, r: [batch][64] (m < N);  c: [batch][128]. */
int mpe_paillier_encrypt(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx,
                         const uint32_t* d_m, const uint32_t* d_r, uint32_t* d_c, void* stream);
/* m = Dec(c), CRT form            `Paillier::decrypt(dk, c)` (src/utilities/mta/mod.rs:165,
 * src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:455-457).  c: [batch][128] -> m: [batch][64].
 * Requires a private key set.  Ciphertexts must be units mod N (as the reference assumes). */
int mpe_paillier_decrypt(mpe_ctx* ctx, const mpe_paillier* sk, int batch, const int32_t* d_key_idx,
                         const uint32_t* d_c, uint32_t* d_m, void* stream);
/* c1*c2 mod N^2                   `Paillier::add` (src/utilities/mta/mod.rs:145) */
int mpe_paillier_add(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx,
                     const uint32_t* d_c1, const uint32_t* d_c2, uint32_t* d_out, void* stream);
/* c^k mod N^2                     `Paillier::mul` (src/utilities/mta/mod.rs:140-144); k: [batch][k_words] */
int mpe_paillier_mul(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx,
                     const uint32_t* d_c, const uint32_t* d_k, int k_words, uint32_t* d_out, void* stream);

/* Kernel geometry chosen for the last launch (for bench.py's roofline accounting). */
typedef struct {
  int waves;              /* workgroups (= waves) launched */
  int ints_per_wave;      /* big integers processed concurrently by one wave */
  int limbs;              /* internal limbs per integer (K) */
  int limb_bits;          /* internal radix (W) */
  int lds_bytes_per_wave;
  size_t table_scratch_bytes;
} mpe_launch_info;
int mpe_last_launch_info(const mpe_ctx* ctx, mpe_launch_info* out);

/* Per-launch timing of the heavy kernels with HIP events recorded on the launch stream (used by
 * bench.py for the roofline line).  kind: 0 = modexp kernel, 1 = modmul kernel. */
typedef struct { int kind; int bits; int exp_words; int batch; float ms; } mpe_prof_rec;
int mpe_prof_enable(mpe_ctx* ctx, int on);       /* clears earlier records */
int mpe_prof_collect(mpe_ctx* ctx, mpe_prof_rec* out, int max_records, int* n_out);  /* waits for the events */

#ifdef __cplusplus
}
#endif
#endif /* MPECDSA_HIP_H */
