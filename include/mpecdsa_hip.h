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
 *  - No mutable global state: distinct mpe_ctx may be used from distinct host threads / streams at the same time
 *    (tests/test_threads_gpu.py).  The environment is read only inside mpe_ctx_create; mpe_last_error() is per thread.
 *
 * Two caveats a binder must know
 *  - RECALLED conventions.  curv-kzen 0.9, kzen-paillier 0.4.2 and zk-paillier 0.4.3 are not vendored with the reference
 *    (Cargo.toml:36-47), so these byte-level details are believed, not read: DigestExt::chain_point hashing the 65-byte
 *    uncompressed point; BigInt::to_bytes(0) = one 0x00 byte; the order of the points in the challenges of DLogProof,
 *    PedersenProof, HomoELGamalProof, ECDDHProof; zk-paillier's SALT_STRING, mask_generation block order and
 *    CompositeDLogProof field order; the serde forms of BigInt / Point / Scalar (host side only: multi_party_ecdsa_amd/wire.py).
 *    Each is a run-time field of `mpe_encoding` (mpe_ctx_set_encoding below), exercised in every alternative by the tests;
 *    tools/diagnose_encodings.py finds the profile of the real crates from one dump of vectors (tools/rust_vectors/run.sh).
 *  - Side channels.  No secret-dependent control flow (fixed window schedules, masked selects; the one exponent whose bits
 *    steer the operation sequence is the PUBLIC key N), but window tables in HBM are indexed by secret exponent windows:
 *    the kernels are NOT constant-time against a co-tenant observing memory-access patterns on the same GPU.  Deploy on a
 *    GPU dedicated to the signing service.  (The reference's mpz_powm makes no constant-time claim either.)
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
/* Zeroes the scratch the context owns (window tables of secret bases, nonce-derived intermediates of composite calls);
 * the reference zeroizes its round-1 secrets (src/utilities/mta/range_proofs.rs:26-36,197-212).  Also done by destroy. */
int mpe_ctx_wipe(mpe_ctx* ctx, void* stream);
/* Audit of the wiping: the number of non-zero 32-bit words in every scratch region the context owns (window tables,
 * composite workspace, cached GG20 session arena and message slabs) and their total size.  Synchronises the stream. */
int mpe_ctx_scratch_audit(mpe_ctx* ctx, uint64_t* nonzero_words, uint64_t* total_bytes, void* stream);
/* A hint for callers that run SEVERAL contexts on one device at the same time (one host thread and stream each).  A single small
 * batch is latency-bound, so by default a context spreads each big integer of a small launch over 2x / 4x the lanes (less efficient,
 * shorter); with `contexts` > 1 the small-batch heuristics compare a launch with 1/contexts of the chip and keep the efficient
 * layouts.  MEASURED (profiles/r04/stream_sweep*.log): the hint LOSES 7-20 % at every depth from 2 to 24 contexts — concurrent small
 * batches are bound by how many kernels run side by side, and the wide layouts fill the chip with fewer of them.  It is kept for
 * experiments only (results are identical whatever the value; default 1).  The supported way to serve a stream of small batches is
 * mpe_gg20_pipeline_* below: one handle, one host thread, batches coalesced per pass. */
int mpe_ctx_set_device_share(mpe_ctx* ctx, int contexts);
/* Run-time options: the A/B switches every "x vs y" figure of DESIGN.md was measured with.  The library reads NO environment
 * variable; a context starts with the shipped defaults and changes only through this call (before the objects that depend on the
 * option are created: e.g. fb_window_bits before mpe_statements_create / mpe_gg20_keys_create).  None changes a result.
 *   0/1 switches   no_fixed_base no_crt no_multiexp no_pair no_pown no_sliding no_par no_wide no_ec_lane_groups
 *                  no_adaptive_lanes (= no_wide + no_ec_lane_groups) no_merge_xn no_merge_r1 gg20_trace
 *                  no_r1_inversion_ahead no_r1_dlog_first (mpe_gg20_sign on small batches: round 0 no longer inverts the ciphertexts
 *                  round 1 needs / MessageB's DLog proofs go behind the ladders again)  no_pdl_ahead (round 4 computes the PDL
 *                  proofs' beta^N itself instead of finding it done)  no_prio (no s_setprio in the ladder kernels)
 *                  no_crt_n (the provers' r^e mod N on the 2048-bit ladder instead of through p | q)
 *   integers       fb_window_bits 4..16 | window_bits 0 (auto), 4..6 | wide_div 1..64 | xwide_div 0 (off).. | waves_per_cu 1..8
 *                  fb_budget_mb | fb_split 0 (auto)..64 | sampler_max_attempts 1.. (default 128)
 *   names          grid = equal | full | hybrid
 * MPE_E_ARG (and mpe_last_error) for an unknown key or a value out of range.  mpe_ctx_get_option returns the integer form;
 * mpe_ctx_option_count / _name enumerate the integer-valued keys. */
int mpe_ctx_set_option(mpe_ctx* ctx, const char* key, const char* value);
int mpe_ctx_get_option(const mpe_ctx* ctx, const char* key, long* value);
int mpe_ctx_option_count(void);
const char* mpe_ctx_option_name(int i);
/* Blocks the host until everything queued on `stream` has finished (hipStreamSynchronize). */
int mpe_sync(mpe_ctx* ctx, void* stream);

/* ---- byte-level conventions of the un-vendored crates -------------------------------------------------------------------
 * The reference pulls curv-kzen 0.9, kzen-paillier 0.4.2 and zk-paillier 0.4.3 from crates.io (Cargo.toml:36-47); none is
 * vendored under /root/reference, so the byte strings their Fiat-Shamir transcripts hash are RECALLED, not read.  Every such
 * convention is a run-time property of a context — never a build-time constant — so that the day a vector dumped from the
 * real crates (tools/rust_vectors/run.sh -> tests/golden/ref_vectors.json) disagrees with a default below, the fix is one
 * mpe_ctx_set_encoding call, not a kernel edit.  tools/diagnose_encodings.py reads such a dump and prints the profile the
 * crates actually use.  The defaults are what this repository believes curv 0.9 does.
 *
 * What is NOT here because the reference's own source fixes it: AliceProof / BobProof / PDLwSlackProof transcripts
 * (range_proofs.rs:143-150,175-182,375-405; zk_pdl_with_slack/mod.rs:102-110: chain_bigint of every field, points as
 * BigInt::from_bytes(to_bytes(true))) and the message of the HashCommitment (party_i.rs:577-580: the compressed point as a
 * BigInt).  They depend on the crates only through `zero_bytes`. */
typedef struct mpe_encoding {
  uint8_t chain_point;      /* curv `DigestExt::chain_point(P)` (DLogProof mta/mod.rs:147-148,170-171; PedersenProof party_i.rs:620-634;
                             * HomoELGamalProof :778-833; ECDDHProof blame.rs:258-320):
                             * 0 = P.to_bytes(false), 65 bytes 04|x|y (default);  1 = P.to_bytes(true), 33 bytes 02/03|x */
  uint8_t zero_bytes;       /* `BigInt::to_bytes()` of the value 0, as hashed by chain_bigint / zk-paillier's compute_digest:
                             * 0 = one 0x00 byte (rust-gmp exports sizeinbase(0) = 1 digit; default);  1 = the empty string */
  uint8_t ck_mask_order;    /* zk-paillier `mask_generation` of NiCorrectKeyProof (party_i.rs:283-301): the 256-bit blocks
                             * H(seed, j), j = 0.. are combined as  0 = sum_j H(seed, j) << (256 j) (default);
                             * 1 = H(seed, 0) || H(seed, 1) || ... (block 0 most significant) */
  uint8_t reserved;         /* must be 0 */
  uint32_t ck_salt;         /* NiCorrectKeyProof's salt as an integer: SALT_STRING = b"KZen" = 0x4B5A656E (default) */
  /* order of the points inside the challenge hash of each curv sigma proof: ord[i] = index (into the canonical list named
   * here) of the point hashed i-th; the default is the identity.  Unused tail entries are ignored. */
  uint8_t ord_dlog[4];      /* DLogProof        (R = pk_t_rand_commitment, G, pk) */
  uint8_t ord_pedersen[8];  /* PedersenProof    (g, h, com, a1, a2) */
  uint8_t ord_heg[8];       /* HomoELGamalProof (T, A3, G, H, Y, D, E) */
  uint8_t ord_ecddh[8];     /* ECDDHProof       (g1, h1, g2, h2, a1, a2) */
  uint8_t ord_cdlog[4];     /* zk-paillier CompositeDLogProof challenge, BigInts (x, g, N, ni)  (party_i.rs:219-258) */
} mpe_encoding;
/* Fills *out with the defaults above.  Never touches the GPU. */
void mpe_encoding_default(mpe_encoding* out);
/* Installs / reads the profile of a context.  set returns MPE_E_ARG unless every flag is 0/1 and every ord_* prefix is a
 * permutation.  Takes effect for every later call on the context (objects created earlier hold no encoded state). */
int mpe_ctx_set_encoding(mpe_ctx* ctx, const mpe_encoding* enc);
int mpe_ctx_get_encoding(const mpe_ctx* ctx, mpe_encoding* out);

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
 * Fixed windows (4/5/6 bits by exponent length), constant sequence of operations for a given exp_words
 * (exponents on this path are secret nonces). */
int mpe_modexp(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx,
               const uint32_t* d_base, const uint32_t* d_exp, int exp_words, uint32_t* d_out,
               void* stream);
/* out[i] = base[i]^exp[i] * base2[i]^exp2[i] mod modulus[idx(i)]: the `mod_pow(..) * mod_pow(..) % n` pattern of
 * the verifiers (range_proofs.rs:134-141, zk_pdl_with_slack/mod.rs:144-157) on ONE ladder — the squarings are
 * shared, the residue is the same.  exp2 must be the short one: 32*exp2_words < 32*exp_words - 6. */
int mpe_modexp2(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx,
                const uint32_t* d_base, const uint32_t* d_exp, int exp_words,
                const uint32_t* d_base2, const uint32_t* d_exp2, int exp2_words, uint32_t* d_out, void* stream);

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
/* kzen-paillier `Open::open(dk, c)` (blame.rs:252-256, `extract_paillier_randomness`): m = Dec(c) and the r with
 * c = (1 + m N) r^N mod N^2, i.e. r = (c mod N)^(N^-1 mod phi(N)) mod N.  d_m, d_r [batch][64]. */
int mpe_paillier_open(mpe_ctx* ctx, const mpe_paillier* sk, int batch, const int32_t* d_key_idx, const uint32_t* d_c,
                      uint32_t* d_m, uint32_t* d_r, void* stream);
int mpe_paillier_add(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx,
                     const uint32_t* d_c1, const uint32_t* d_c2, uint32_t* d_out, void* stream);
/* c^k mod N^2                     `Paillier::mul` (src/utilities/mta/mod.rs:140-144); k: [batch][k_words] */
int mpe_paillier_mul(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx,
                     const uint32_t* d_c, const uint32_t* d_k, int k_words, uint32_t* d_out, void* stream);

/* ---- BigInt::mod_inv ------------------------------------------------------------------------ */
/* out[i] = a[i]^-1 mod modulus[idx(i)], ok[i] = 0 when gcd != 1 (curv `BigInt::mod_inv` -> Option;
 * src/utilities/mta/range_proofs.rs:122,135,339,351,363; zk_pdl_with_slack/mod.rs:192).
 * a must be reduced (a < modulus). */
int mpe_modinv(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx, const uint32_t* d_a,
               uint32_t* d_out, uint8_t* d_ok, void* stream);

/* ---- secp256k1 (curv Point<Secp256k1> / Scalar<Secp256k1>) -------------------------------------- */
/* Points: 16 words, affine x[8] | y[8], all-zero = point at infinity.  Scalars: k_words little-endian
 * words, reduced mod q on entry exactly like `Scalar::from(&BigInt)` (k_words <= 89).
 * out = k*G (`Point::generator() * k`), out = k*P (`P * k`), out = P + Q. */
int mpe_ec_mul_base(mpe_ctx* ctx, int batch, const uint32_t* d_k, int k_words, uint32_t* d_out, void* stream);
int mpe_ec_mul(mpe_ctx* ctx, int batch, const uint32_t* d_k, int k_words, const uint32_t* d_P, uint32_t* d_out,
               void* stream);
int mpe_ec_add(mpe_ctx* ctx, int batch, const uint32_t* d_P, const uint32_t* d_Q, uint32_t* d_out, void* stream);

/* curv `DLogProof::prove(sk)` with the nonce as input / `DLogProof::verify`
 * (src/utilities/mta/mod.rs:147-148,170-171).  pk, R: points; z: 8 words. */
int mpe_dlog_prove(mpe_ctx* ctx, int batch, const uint32_t* d_sk, const uint32_t* d_nonce, uint32_t* d_pk,
                   uint32_t* d_R, uint32_t* d_z, void* stream);
int mpe_dlog_verify(mpe_ctx* ctx, int batch, const uint32_t* d_pk, const uint32_t* d_R, const uint32_t* d_z,
                    uint8_t* d_ok, void* stream);

/* ---- DLogStatement tables and the range / PDL proofs -------------------------------------------- */
/* `zk_paillier::DLogStatement{N, g, ni}` as GG20 stores (N~, h1, h2)
 * (src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:225-229).  d_Nt, d_h1, d_h2: [count][64]. */
typedef struct mpe_statements mpe_statements;
int mpe_statements_create(mpe_ctx* ctx, int count, const uint32_t* d_Nt, const uint32_t* d_h1,
                          const uint32_t* d_h2, mpe_statements** out, void* stream);
/* the same with an explicit window width (bits, 2..16) of the fixed-base tables of h1, h2: 2 * count tables of
 * ceil(2848 / wb) * 2^wb rows of 288 bytes (mpe_statements_create uses 13: 0.5 GB per base); wb = 0: no tables
 * (statements used once, e.g. the fresh (N~, h1, h2) of a Lindell'17 key generation) */
int mpe_statements_create_wb(mpe_ctx* ctx, int count, const uint32_t* d_Nt, const uint32_t* d_h1, const uint32_t* d_h2,
                             int wb, mpe_statements** out, void* stream);
int mpe_statements_destroy(mpe_statements* s);

/* Field widths (words): z,s 64 | e 8 | s1 25 (< 2^769) | s2,s3 89 (< 2^2817) | u2 128 | points 16.
 * Nonces (the reference samples them inside the call; here they are inputs):
 * alpha 24 (< q^3) | beta 64 | gamma 88 (< q^3 N~) | rho 72 (< q N~). */
typedef struct { uint32_t *z, *e, *s, *s1, *s2; } mpe_alice_proof;
typedef struct { const uint32_t *alpha, *beta, *gamma, *rho; } mpe_alice_nonces;
/* `AliceProof::generate(a, cipher, alice_ek, dlog_statement, r)`  (src/utilities/mta/range_proofs.rs:160-193)
 * a: [batch][8], cipher: [batch][128], r: [batch][64] (the Paillier randomness of cipher). */
int mpe_alice_generate(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                       const int32_t* d_key_idx, const int32_t* d_st_idx, const uint32_t* d_a,
                       const uint32_t* d_cipher, const uint32_t* d_r, const mpe_alice_nonces* nonces,
                       const mpe_alice_proof* out, void* stream);
/* `AliceProof::verify(cipher, alice_ek, dlog_statement) -> bool`  (range_proofs.rs:105-156) */
int mpe_alice_verify(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                     const int32_t* d_key_idx, const int32_t* d_st_idx, const uint32_t* d_cipher,
                     const mpe_alice_proof* proof, uint8_t* d_ok, void* stream);

typedef struct { uint32_t *z, *u1, *u2, *u3, *s1, *s2, *s3; } mpe_pdl_proof;
typedef struct { const uint32_t *alpha, *beta, *rho, *gamma; } mpe_pdl_nonces;
/* `PDLwSlackProof::prove(witness{x, r}, statement{ciphertext, ek, Q, G, h1, h2, N_tilde})`
 * (src/utilities/zk_pdl_with_slack/mod.rs:68-125).  Q, G: points; x: [batch][8]; r: [batch][64]. */
int mpe_pdl_prove(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                  const int32_t* d_key_idx, const int32_t* d_st_idx, const uint32_t* d_cipher,
                  const uint32_t* d_Q, const uint32_t* d_G, const uint32_t* d_x, const uint32_t* d_r,
                  const mpe_pdl_nonces* nonces, const mpe_pdl_proof* out, void* stream);
/* `PDLwSlackProof::verify(statement) -> Result<(), ZkPdlWithSlackError>`  (mod.rs:127-179); ok[i]=1 accepts.
 * (Where the reference would panic on a non-invertible c or z, the item is rejected instead.) */
int mpe_pdl_verify(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                   const int32_t* d_key_idx, const int32_t* d_st_idx, const uint32_t* d_cipher,
                   const uint32_t* d_Q, const uint32_t* d_G, const mpe_pdl_proof* proof, uint8_t* d_ok,
                   void* stream);

/* ---- MtA share conversion (src/utilities/mta/mod.rs) ---------------------------------------------------- */
/* `dlog_statements` = all statements of `stm` (rounds.rs:87,154 pass the whole h1_h2_n_tilde_vec); per-exchange
 * arrays of range proofs / nonces are item-major [batch][count_statements]. */
typedef struct { uint32_t *pk, *R, *z; } mpe_dlog_proof;     /* DLogProof{pk, pk_t_rand_commitment, challenge_response} */
/* `MessageA::a_with_predefined_randomness(a, alice_ek, randomness, dlog_statements)`  (:62-87)
 * a [batch][8], r [batch][64] -> c [batch][128] + one AliceProof per statement. */
int mpe_mta_message_a(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                      const int32_t* d_key_idx, const uint32_t* d_a, const uint32_t* d_r,
                      const mpe_alice_nonces* nonces, uint32_t* d_c, const mpe_alice_proof* proofs, void* stream);
/* `MessageB::b_with_predefined_randomness(b, alice_ek, m_a, randomness, beta_tag, dlog_statements)`  (:111-158)
 * key_idx selects ALICE's key.  ok[i] = 0 is the reference's Err(InvalidKey) (a range proof failed); the
 * outputs of such an item are still written.  beta = -beta_tag mod q [batch][8]. */
int mpe_mta_message_b(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                      const int32_t* d_key_idx, const uint32_t* d_b, const uint32_t* d_ca,
                      const mpe_alice_proof* range_proofs, const uint32_t* d_r, const uint32_t* d_beta_tag,
                      const uint32_t* d_nonce_b, const uint32_t* d_nonce_bt, uint32_t* d_cb, uint32_t* d_beta,
                      const mpe_dlog_proof* b_proof, const mpe_dlog_proof* beta_tag_proof, uint8_t* d_ok, void* stream);
/* `MessageB::verify_proofs_get_alpha(dk, a)`  (:160-179): alpha [batch][8] = Dec(c_b) mod q, alice_share [batch][64]
 * the full plaintext; ok[i] = 0 is Err(InvalidKey). */
int mpe_mta_verify_get_alpha(mpe_ctx* ctx, const mpe_paillier* sk, int batch, const int32_t* d_key_idx,
                             const uint32_t* d_cb, const mpe_dlog_proof* b_proof, const mpe_dlog_proof* beta_tag_proof,
                             const uint32_t* d_a, uint32_t* d_alpha, uint32_t* d_alice_share, uint8_t* d_ok, void* stream);

/* Bob's MtA(wc) range proof: `BobProof::generate` / `BobProof::verify` / `BobProofExt::verify`
 * (src/utilities/mta/range_proofs.rs:218-534).  Widths (words): t,z,s 64 | e 8 | s1 25 | s2,t2 89 | t1 81 (< 2^2561).
 * Nonces: alpha 24 (< q^3) | beta 64 | gamma 80 (< q^2 N) | rho, sigma 72 (< q N~) | rho_prim, tau 88 (< q^3 N~).
 * generate: a_enc, mta_enc [batch][128]; b [batch][8]; beta_prim, r [batch][64]; check != 0 also writes
 * u = alpha*G to d_u [batch][16] and hashes X = b*G, u (the `check` flag of :414-424).
 * verify: d_X, d_u both NULL -> BobProof::verify(.., None); both set -> BobProofExt::verify. */
typedef struct { uint32_t *t, *z, *e, *s, *s1, *s2, *t1, *t2; } mpe_bob_proof;
typedef struct { const uint32_t *alpha, *beta, *gamma, *rho, *rho_prim, *sigma, *tau; } mpe_bob_nonces;
int mpe_bob_generate(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                     const int32_t* d_key_idx, const int32_t* d_st_idx, const uint32_t* d_a_enc,
                     const uint32_t* d_mta_enc, const uint32_t* d_b, const uint32_t* d_beta_prim, const uint32_t* d_r,
                     const mpe_bob_nonces* nonces, int check, const mpe_bob_proof* out, uint32_t* d_u, void* stream);
int mpe_bob_verify(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                   const int32_t* d_key_idx, const int32_t* d_st_idx, const uint32_t* d_a_enc,
                   const uint32_t* d_mta_enc, const mpe_bob_proof* proof, const uint32_t* d_X, const uint32_t* d_u,
                   uint8_t* d_ok, void* stream);

/* ---- curv sigma proofs of GG20 phases 3 / 6 and the phase-1 hash commitment (curv-kzen 0.9, un-vendored) ------ */
/* `PedersenProof::prove(m, r)` with the nonces s1, s2 as inputs (src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:620-634):
 * com = m G + r H (H = `Point::base_point2()`), a1 = s1 G, a2 = s2 H, e = H(G, H, com, a1, a2), z1 = s1 + e m, z2 = s2 + e r.
 * Points 16 words, scalars 8.  `PedersenProof::verify` (state_machine/sign/rounds.rs:371-378) recomputes e. */
typedef struct { uint32_t *com, *e, *a1, *a2, *z1, *z2; } mpe_pedersen_proof;
int mpe_pedersen_prove(mpe_ctx* ctx, int batch, const uint32_t* d_m, const uint32_t* d_r, const uint32_t* d_s1,
                       const uint32_t* d_s2, const mpe_pedersen_proof* out, void* stream);
int mpe_pedersen_verify(mpe_ctx* ctx, int batch, const mpe_pedersen_proof* proof, uint8_t* d_ok, void* stream);
/* `HomoELGamalProof::prove(witness{x, r}, statement{G, H, Y, D, E})` / `verify` (party_i.rs:778-833): T = s1 H + s2 Y,
 * A3 = s2 G, e = H(T, A3, G, H, Y, D, E), z1 = s1 + e x (s1 when x = 0), z2 = s2 + e r;
 * verify: z1 H + z2 Y == T + e D and z2 G == A3 + e E.  Every point per item, [batch][16]. */
typedef struct { const uint32_t *G, *H, *Y, *D, *E; } mpe_heg_statement;
typedef struct { uint32_t *T, *A3, *z1, *z2; } mpe_heg_proof;
int mpe_heg_prove(mpe_ctx* ctx, int batch, const uint32_t* d_x, const uint32_t* d_r, const uint32_t* d_s1, const uint32_t* d_s2,
                  const mpe_heg_statement* statement, const mpe_heg_proof* out, void* stream);
int mpe_heg_verify(mpe_ctx* ctx, int batch, const mpe_heg_statement* statement, const mpe_heg_proof* proof, uint8_t* d_ok,
                   void* stream);
/* `HashCommitment::create_commitment_with_user_defined_randomness(BigInt::from_bytes(P.to_bytes(true)), blind)`
 * (party_i.rs:577-580,654-659): com [batch][8] = SHA-256(33 bytes of P || minimal bytes of blind). */
int mpe_hash_commit_point(mpe_ctx* ctx, int batch, const uint32_t* d_P, const uint32_t* d_blind, uint32_t* d_com, void* stream);

/* ---- GG20 signing ------------------------------------------------------------------------------------------------ */
/* Key material as the reference's keygen leaves it in `LocalKey` (state_machine/keygen/rounds.rs:311-322), for
 * `nkeysets` wallets of the same (t, n) shape (a batch may mix wallets: every session names its key set).
 * PUBLIC part, all n parties:  d_N [K][n][64] `paillier_key_vec`, d_Nt/d_h1/d_h2 [K][n][64] `h1_h2_n_tilde_vec`,
 * d_y [K][16] `y_sum_s`, d_X [K][n][16] `pk_vec`.
 * SECRET part, only of the n_own parties h_own[] (ascending party indices) this process acts for:
 * d_x [K][n_own][8] `keys_linear.x_i`, d_p / d_q [K][n_own][32] `paillier_dk`.  One party per process is the
 * reference's deployment; all of them in one object is its `Simulation` test harness (state_machine/sign.rs:667-763).
 * h_signers (HOST array): n_signers ascending party indices (`s_l` minus one, sign.rs:78), t < n_signers <= n.
 * The window width of the fixed-base tables of h1, h2 is chosen from the table memory 2 K n bases need
 * (13 bits = 0.5 GB per base when that fits a quarter of the free HBM, narrower for many wallets). */
typedef struct mpe_gg20_keys mpe_gg20_keys;
int mpe_gg20_keys_create(mpe_ctx* ctx, int t, int n, int n_signers, const int32_t* h_signers, int nkeysets, int n_own,
                         const int32_t* h_own, const uint32_t* d_x, const uint32_t* d_p, const uint32_t* d_q,
                         const uint32_t* d_N, const uint32_t* d_Nt, const uint32_t* d_h1, const uint32_t* d_h2,
                         const uint32_t* d_y, const uint32_t* d_X, mpe_gg20_keys** out, void* stream);
int mpe_gg20_keys_destroy(mpe_gg20_keys* keys);
int mpe_gg20_keys_fb_window_bits(const mpe_gg20_keys* keys);

/* Everything the reference samples while signing, as inputs.  L = the local parties of a session object (their
 * nonces only), S = signers.  Leading dimensions [B][L]; then statement st (n); peer slot jj (S-1; the peer's signer
 * ordinal is ind = jj < i ? jj : jj+1, the `ind` of rounds.rs:149); MessageB variant v (0: gamma_i, 1: w_i).
 *   k, gamma, blind, l, ped_s1, ped_s2, heg_s1, heg_s2 : [B][L][8]      r_a : [B][L][64]
 *   al_alpha [B][L][n][24]  al_beta [..][64]  al_gamma [..][88]  al_rho [..][72]       (AliceProof nonces)
 *   mb_beta_tag, mb_r : [B][L][S-1][2][64]   mb_nonce_b, mb_nonce_bt : [B][L][S-1][2][8] (MessageB::b)
 *   pdl_alpha [B][L][S-1][24]  pdl_beta [..][64]  pdl_rho [..][72]  pdl_gamma [..][88]   (PDLwSlackProof::prove)
 *   msg : [B][8]  the message as BigInt (reduced mod q like `Scalar::from(message)`, party_i.rs:857); read by
 *         mpe_gg20_sign only (the round view takes the message in round 7). */
typedef struct {
  const uint32_t *k, *gamma, *blind, *r_a;
  const uint32_t *al_alpha, *al_beta, *al_gamma, *al_rho;
  const uint32_t *mb_beta_tag, *mb_r, *mb_nonce_b, *mb_nonce_bt;
  const uint32_t *l, *ped_s1, *ped_s2;
  const uint32_t *pdl_alpha, *pdl_beta, *pdl_rho, *pdl_gamma;
  const uint32_t *heg_s1, *heg_s2;
  const uint32_t *msg;
} mpe_gg20_nonces;

/* ---- the SAMPLING side of the trait surface ------------------------------------------------------------------------
 * curv `Samplable::{sample, sample_below, sample_range}`, the reference's `SampleFromMultiplicativeGroup::from_modulo`
 * (src/utilities/mta/range_proofs.rs:538-557) and `Scalar::<Secp256k1>::random()`, drawn ON THE DEVICE from a 32-byte seed.
 * Item g of stream `stream_id` reads the ChaCha20 keystream (RFC 8439 block function) with key = seed, state[12] = block
 * counter from 0, state[13] = g, state[14] | state[15] << 32 = stream_id, taking its bytes in order; the byte -> integer
 * rules are curv's (recalled): sample(bits) = from_bytes_be(ceil(bits/8) bytes) >> (8 ceil(bits/8) - bits);
 * sample_below(u) = repeat sample(bit_length(u)) until < u (every attempt takes fresh bytes).
 * A (seed, stream_id) pair must never be used twice.  h_seed32: HOST pointer to 32 bytes.
 *   mpe_sample_bits    d_out [batch][out_words] = BigInt::sample(bits)
 *   mpe_sample_below   d_out = sample_below(bound row d_bound_idx[i] of d_bound [nbounds][bound_words]; NULL index: row 0 when
 *                      nbounds == 1, else row i); flags: MPE_SAMPLE_NONZERO rejects 0, MPE_SAMPLE_PLUS_ONE returns 1 + the draw
 *                      (sample_range(1, bound + 1)), MPE_SAMPLE_COPRIME = from_modulo: also repeat until gcd(x, bound) == 1
 *                      (odd bounds of at most 2048 bits; an even bound counts as a failure)
 *   mpe_sample_scalar  d_out [batch][8] = Scalar::random(): 32 bytes as a big-endian integer until 0 < x < q
 * d_fail (may be NULL): incremented for every item that exhausted 128 attempts (probability < 2^-128; its row is zero). */
#define MPE_SAMPLE_NONZERO 1
#define MPE_SAMPLE_PLUS_ONE 2
#define MPE_SAMPLE_COPRIME 4
int mpe_sample_bits(mpe_ctx* ctx, int batch, const uint8_t* h_seed32, uint64_t stream_id, int bits, int out_words, uint32_t* d_out,
                    void* stream);
int mpe_sample_below(mpe_ctx* ctx, int batch, const uint8_t* h_seed32, uint64_t stream_id, const uint32_t* d_bound, int bound_words,
                     int nbounds, const int32_t* d_bound_idx, int flags, int out_words, uint32_t* d_out, int32_t* d_fail, void* stream);
int mpe_sample_scalar(mpe_ctx* ctx, int batch, const uint8_t* h_seed32, uint64_t stream_id, uint32_t* d_out, int32_t* d_fail,
                      void* stream);
/* Everything a batch of signing sessions draws from OsRng, with the reference's distributions, into the arrays of `out`
 * (all fields but msg; the layout of mpe_gg20_nonces for `n_local` local parties h_local[]):
 *   k, gamma, l, ped_s*, heg_s*, mb_nonce_*: Scalar::random() (party_i.rs:561-563,628; curv's sigma proofs; mta/mod.rs:147-148)
 *   blind: BigInt::sample(256) (party_i.rs:574)     r_a: sample_below(N_i) (mta/mod.rs:57)
 *   al_alpha < q^3, al_beta = from_modulo(N_i), al_gamma < q^3 N~_st, al_rho < q N~_st   (range_proofs.rs:48-51)
 *   mb_beta_tag, mb_r < N_peer (mta/mod.rs:97-98)
 *   pdl_alpha < q^3, pdl_beta = sample_range(1, N_i - 1), pdl_rho < q N~_peer, pdl_gamma < q^3 N~_peer (zk_pdl_with_slack/mod.rs:73-77)
 * field f (its position in mpe_gg20_nonces, k = 0) of batch `batch_counter` (< 2^56) uses stream_id = batch_counter | f << 56 and
 * item index = the field's row index.  from_modulo's gcd runs as ONE batched verdict (Montgomery's trick) and only failing
 * items are redrawn lane by lane — the values equal those of the literal loop.  A (seed, batch_counter) pair signs ONE batch.
 * mpe_gg20_nonces_alloc / _view / _free: one device allocation holding every field for (batch, n_local), zeroed when freed. */
typedef struct mpe_gg20_nonce_buf mpe_gg20_nonce_buf;
int mpe_gg20_nonces_alloc(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, int n_local, mpe_gg20_nonce_buf** out);
int mpe_gg20_nonces_view(const mpe_gg20_nonce_buf* buf, mpe_gg20_nonces* out);
int mpe_gg20_nonces_free(mpe_gg20_nonce_buf* buf);
int mpe_gg20_sample_nonces(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, int n_local, const int32_t* h_local,
                           const int32_t* d_keyset, const uint8_t* h_seed32, uint64_t batch_counter, const mpe_gg20_nonces* out,
                           int32_t* d_fail, void* stream);

/* GG20 round messages.  One fixed-size record of 32-bit words per (sender, session); a slab holds, for every sender,
 * a [B][W] block of records.  P2P messages travel like broadcast ones and are filtered by the receiver, exactly as the
 * reference's relay does (examples/gg20_sm_client.rs:35-40).  Points: x[8] | y[8]; all other fields little-endian words.
 *   round 0  (MessageA, SignBroadcastPhase1)       W = 256 (n+1): sub-record st < n = AliceProof for statement st
 *                                                   {z 0, e 64, s 72, s1 136, s2 161}; sub-record n = {c 0, com 128}
 *   round 1  (GammaI, WI) to every peer            W = 208 * 2 (S-1): sub-record 2 jj + v = MessageB for peer slot jj
 *                                                   {c 0, b_proof {pk 128, R 144, z 160}, beta_tag_proof {pk 168, R 184, z 200}}
 *   round 2  (DeltaI, TI, TIProof)                 W = 96: {delta 0, T 8, proof {e 24, a1 32, a2 48, com 64, z1 80, z2 88}}
 *   round 3  SignDecommitPhase1                    W = 24: {blind_factor 0, g_gamma_i 8}
 *   round 4  (RDash, Vec<PDLwSlackProof>)          W = 450 S: sub-record jj < S-1 = the proof for peer slot jj
 *                                                   {z 0, u1 64, u2 80, u3 208, s1 272, s2 297, s3 361}; sub-record S-1 = {R_dash 0}
 *   round 5  (SI, HEGProof)                        W = 64: {S_i 0, T 16, A3 32, z1 48, z2 56}
 *   round 7  PartialSignature                      W = 8:  {s_i}
 * mpe_gg20_msg_words returns W (0 for rounds that emit nothing). */
int mpe_gg20_msg_words(int n_signers, int n, int round);

/* The per-party ROUND VIEW: `RoundN::proceed` (state_machine/sign/rounds.rs:68,122,234,347,431,525,612,672) batched over
 * `batch` sessions and over the n_local parties h_local[] (ascending signer ORDINALS, positions in s_l) this object acts
 * for; their secrets must be in `keys`.  d_keyset [batch]: key set of every session (NULL when nkeysets == 1).
 * `nonces`: the local parties' sampled values (layout above); the arrays must stay valid until round 5 is queued.
 * dedup_verify = 0: faithful work (each range proof verified for both MessageB::b calls, every local party verifies
 * every PDL proof, rounds.rs:151-175,546-558); 1: identical checks are evaluated once (same results).
 *
 * mpe_gg20_roundN(sess, d_in, h_in_off, d_out): d_in = the previous round's records of ALL S senders ("including me"):
 * sender ordinal j's [batch][W] block starts at record h_in_off[j] of d_in (HOST array; NULL = j*batch); d_out = this
 * object's outgoing records [n_local][batch][W].  Round 6 emits nothing; round 7 takes the messages to sign
 * (d_msg [batch][8]) and emits the partial signatures; mpe_gg20_complete = `SignManual::complete` (sign.rs:625-646).
 * Rounds must be called in order.  A failing check never aborts the batch: the party's status in that session becomes
 * 100*round + detail of its FIRST failed check, in the order the reference evaluates them, and sticks:
 *   101 MessageB::b -> InvalidKey (Error::Round1) | 201 verify_proofs_get_alpha | 202 b_proof.pk != g_w_vec[ind] (rounds.rs:281)
 *   303 T_i != proof.com (rounds.rs:366) | 301 delta not invertible | 302 PedersenProof::verify
 *   401 phase4 "bad gamma_i decommit" | 501 "Bad PDLwSlack proof" | 502 phase5_check_R_dash_sum
 *   601 phase6_verify_proof | 602 phase6_check_S_i_sum | 701 output_signature: verify failed
 *   91 (MPE_GG20_STATUS_BAD_NONCE, round 0) k_i or gamma_i is not a value `Scalar::random()` returns (0 < x < q): the device sampler
 *      marks the party of a session one of whose rejection loops gave up this way (mpe_gg20_sample_nonces: after
 *      sampler_max_attempts candidates, default 128 — probability < 2^-128 per draw; curv's loops are unbounded, a deliberate
 *      divergence) so that a session NEVER signs on zeroed values; a caller's own out-of-range k_i / gamma_i is refused alike
 * with `bad_actors` (a bit mask over signer ordinals, the reference's `ErrorType::bad_actors`, gg_2020/mod.rs:23-27) set
 * for 401 (the peers whose decommitment is bad), 501 (the first failing prover) and 601 (every failing prover).
 * mpe_gg20_session_result: d_status, d_bad_actors, d_recid [n_local][batch]; d_r, d_s [n_local][batch][8] (after
 * mpe_gg20_complete; zero unless status == 0); d_R [n_local][batch][16] (after round 4).  Any pointer may be NULL.
 * Destroying a session zeroes its state (k_i, gamma_i, w_i, sigma_i, ...). */
#define MPE_GG20_STATUS_BAD_NONCE 91
typedef struct mpe_gg20_session mpe_gg20_session;
int mpe_gg20_session_create(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, int n_local, const int32_t* h_local,
                            const int32_t* d_keyset, const mpe_gg20_nonces* nonces, int dedup_verify, mpe_gg20_session** out,
                            void* stream);
int mpe_gg20_session_destroy(mpe_gg20_session* sess, void* stream);
/* The next batch of the same shape on the same object (a party process signs batch after batch: `SignManual::new` again
 * with fresh `SignKeys`, sign.rs:540-569): new sampled values (and key-set choice), state of the previous batch zeroed, rounds
 * start again at 0.  Results are identical to those of a freshly created session.
 * `nonces` MUST hold freshly sampled values (the same buffers refilled are fine): signing two batches with the same k_i,
 * gamma_i or Paillier randomness leaks the key share, and the library cannot detect it.  MPE_E_ARG when the previous batch is
 * half-way through the protocol and healthy (finish it with mpe_gg20_complete, or give it up with mpe_gg20_session_abort);
 * MPE_E_HIP when the wipe fails. */
int mpe_gg20_session_rearm(mpe_gg20_session* sess, const int32_t* d_keyset, const mpe_gg20_nonces* nonces, void* stream);
/* Gives the running batch up (the caller stops after the offline stage, a peer vanished, a round call returned an error): the
 * nonce-derived state is wiped at once, every round entry point refuses, and mpe_gg20_session_rearm starts the next batch.  A
 * session whose round call FAILED (MPE_E_NOMEM, MPE_E_HIP) may also be re-armed directly: a long-lived party process is never
 * left with an object it can only destroy. */
int mpe_gg20_session_abort(mpe_gg20_session* sess, void* stream);
int mpe_gg20_round0(mpe_gg20_session* sess, uint32_t* d_out, void* stream);
int mpe_gg20_round1(mpe_gg20_session* sess, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream);
int mpe_gg20_round2(mpe_gg20_session* sess, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream);
int mpe_gg20_round3(mpe_gg20_session* sess, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream);
int mpe_gg20_round4(mpe_gg20_session* sess, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream);
int mpe_gg20_round5(mpe_gg20_session* sess, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream);
int mpe_gg20_round6(mpe_gg20_session* sess, const uint32_t* d_in, const int64_t* h_in_off, void* stream);
int mpe_gg20_round7(mpe_gg20_session* sess, const uint32_t* d_msg, uint32_t* d_out, void* stream);
int mpe_gg20_complete(mpe_gg20_session* sess, const uint32_t* d_in, const int64_t* h_in_off, void* stream);
int mpe_gg20_session_result(const mpe_gg20_session* sess, int32_t* d_status, uint32_t* d_bad_actors, uint32_t* d_r,
                            uint32_t* d_s, int32_t* d_recid, uint32_t* d_R, void* stream);

/* Fault injection of the reference's own tests (gg_2020/test.rs:282-289,458-465,679-686): the local parties whose signer
 * ordinal is in party_mask double their delta_i (step 5), sigma_i (step 6) or s_i (step 7).  step 0 switches it off. */
int mpe_gg20_session_fault_inject(mpe_gg20_session* sess, int step, uint32_t party_mask);

/* ---- identifiable abort: src/protocols/multi_party_ecdsa/gg_2020/blame.rs ---------------------------------------- */
/* Every signer has opened the values the failing phase used; the functions re-derive the public ciphertexts from the
 * openings under the signers' PUBLIC Paillier keys and name the parties whose openings do not match or whose broadcast
 * value is inconsistent.  Layout: leading dimensions [batch][S] (signer ordinal), then the peer slot j (S-1; the peer's
 * ordinal is ind = j < i ? j : j+1).  d_bad_actors [batch]: bit mask over signer ordinals = the reference's sorted,
 * de-duplicated `ErrorType::bad_actors`.
 * `GlobalStatePhase5::phase5_blame` (blame.rs:116-224), called when phase5_check_R_dash_sum failed (status 502):
 *   k, gamma [B][S][8]; k_rand [B][S][64] (the MessageA randomness); beta_tag, beta_rand [B][S][S-1][64] (for Alice i, slot j:
 *   what Bob `ind` used in his gamma MessageB to i); delta [B][S][8]; g_gamma [B][S][16]; c_a [B][S][128] (m_a_vec[i].c);
 *   c_b [B][S][S-1][128] (m_b_mat[i][j].c, the gamma MessageB ciphertexts Alice i received). */
typedef struct { const uint32_t *k, *k_rand, *gamma, *beta_tag, *beta_rand, *delta, *g_gamma, *c_a, *c_b; } mpe_gg20_blame5_in;
int mpe_gg20_blame5(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, const int32_t* d_keyset, const mpe_gg20_blame5_in* in,
                    uint32_t* d_bad_actors, void* stream);
/* `GlobalStatePhase6::phase6_blame` (blame.rs:322-421), called when phase6_check_S_i_sum failed (status 602):
 *   k, k_rand as above; miu, miu_rand [B][S][S-1][64] (plaintext before reduction and Paillier randomness of the w_i MessageB
 *   ciphertexts Alice i received, `Paillier::open`); a1, a2 [B][S][16], z [B][S][8] (each signer's ECDDHProof that
 *   S_i = sigma_i R); S [B][S][16]; c_a [B][S][128]; c_b [B][S][S-1][128] (the w_i MessageB ciphertexts); R [B][16]. */
typedef struct { const uint32_t *k, *k_rand, *miu, *miu_rand, *a1, *a2, *z, *S, *c_a, *c_b, *R; } mpe_gg20_blame6_in;
int mpe_gg20_blame6(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, const int32_t* d_keyset, const mpe_gg20_blame6_in* in,
                    uint32_t* d_bad_actors, void* stream);
/* `GlobalStatePhase7::phase7_blame` (blame.rs:434-454), called when the signature did not verify (status 701):
 *   s [B][S][8] (partial signatures); r, m [B][8]; R_dash, S [B][S][16]; R [B][16]:  R s_i == m R_dash_i + r S_i. */
typedef struct { const uint32_t *s, *r, *R_dash, *m, *R, *S; } mpe_gg20_blame7_in;
int mpe_gg20_blame7(mpe_ctx* ctx, int n_signers, int batch, const mpe_gg20_blame7_in* in, uint32_t* d_bad_actors, void* stream);
/* What a local party publishes for the phase-6 blame beyond the inputs it was given (LocalStatePhase6, blame.rs:227-234):
 * d_miu [n_local][batch][S-1][64] and the ECDDH proof d_a1, d_a2 [n_local][batch][16], d_z [n_local][batch][8] that
 * S_i = sigma_i R (`GlobalStatePhase6::ecddh_proof`, blame.rs:258-272; d_nonce [batch][n_local][8] is its sampled value).
 * Valid after round 5.  (The Paillier randomness of the incoming ciphertexts: mpe_paillier_open.) */
int mpe_gg20_session_blame6_state(const mpe_gg20_session* sess, const uint32_t* d_nonce, uint32_t* d_miu, uint32_t* d_a1,
                                  uint32_t* d_a2, uint32_t* d_z, void* stream);
/* curv `ECDDHProof::{prove, verify}` for the statement {g1, h1 = x g1, g2, h2 = x g2}: a1 = s g1, a2 = s g2,
 * e = H(g1, h1, g2, h2, a1, a2), z = s + e x; verify: z g1 == a1 + e h1 and z g2 == a2 + e h2.  Points [batch][16]. */
typedef struct { const uint32_t *g1, *h1, *g2, *h2; } mpe_ecddh_statement;
typedef struct { uint32_t *a1, *a2, *z; } mpe_ecddh_proof;
int mpe_ecddh_prove(mpe_ctx* ctx, int batch, const uint32_t* d_x, const uint32_t* d_s, const mpe_ecddh_statement* statement,
                    const mpe_ecddh_proof* out, void* stream);
int mpe_ecddh_verify(mpe_ctx* ctx, int batch, const mpe_ecddh_statement* statement, const mpe_ecddh_proof* proof, uint8_t* d_ok,
                     void* stream);

/* The lock-step composition of the rounds above with every signer local (needs every signer's secrets in `keys`):
 * OfflineStage Round0..Round6 and SignManual for `batch` sessions on this GPU — what `round_based::dev::Simulation`
 * does for one session (state_machine/sign.rs:667-763).  nonces: [batch][S] layout.  Outputs per session: d_r, d_s
 * [batch][8] and d_recid [batch] = `SignatureRecid{r,s,recid}` (party_i.rs:131-135, low-s normalised, 873-910; zero
 * unless status == 0), optionally d_R [batch][16], and d_status [batch] = the smallest non-zero party status (0 = every
 * party produced and verified the signature).  chunk: sessions per internal pass (0 = 65536, fewer for wide shapes so
 * that a pass stays under ~64 GB).  Scratch that held nonce-derived values is zeroed before the call returns. */
int mpe_gg20_sign(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, const int32_t* d_keyset, const mpe_gg20_nonces* nonces,
                  uint32_t* d_r, uint32_t* d_s, int32_t* d_recid, uint32_t* d_R, int32_t* d_status, int dedup_verify, int chunk,
                  void* stream);

/* ---- a stream of batches: the pipelined engine ------------------------------------------------------------------------
 * A signing service receives batch after batch of `batch` sessions (BASELINE config 4: 1 024) and one such batch alone is
 * latency-bound (chains of ~2 048 dependent squarings at a fraction of the chip).  The reference runs many `OfflineStage`s
 * concurrently on one executor (state_machine/sign.rs:667-691; rounds.rs:106,215,323 `is_expensive`).  Here:
 * `group` consecutive batches are coalesced into ONE lock-step pass (every heavy launch carries the items of all of them) and
 * `lanes` (1..4) such passes are in flight at once, each on one stream of its own with its own workspace — at most 4 streams,
 * the runtime's default hardware queues, ONE host thread: submit only enqueues.  Results are bit-identical to mpe_gg20_sign.
 *   submit         the caller's sampled values of ONE batch (layout of mpe_gg20_nonces with every signer local, as for
 *                  mpe_gg20_sign) and where its results go: d_r, d_s [batch][8], d_recid, d_status [batch], d_R [batch][16] or NULL.
 *                  `stream`: the stream that produced the inputs (the pipeline waits for it).  Inputs are copied and results
 *                  written ASYNCHRONOUSLY: both sets of arrays must stay valid until the ticket completes.
 *   submit_seeded  the same, but every sampled value is drawn on the device from (h_seed32, batch_counter) exactly as
 *                  mpe_gg20_sample_nonces does (one sampler launch per group, right before its pass; group <= 16); d_msg [batch][8]
 *                  are the messages.  A (seed, batch_counter) pair signs ONE batch.  A group holds batches of one form only.
 *   flush          sends the partly filled group to the device (a service calls it when its queue runs dry)
 *   wait           blocks the host until the batch of `ticket` is complete (flushing its group if it is still open);
 *   stream_wait    makes `stream` wait for it instead;  query: *done = 0 / 1 without blocking
 *   latency_ms     device time from the submit call to the completion of that batch's results (includes the time the batch waited
 *                  for its group to fill and for a lane);  pass_ms: from the start of the pass that carried it to its completion
 * One pipeline object is driven by one host thread at a time.  Destroying it waits for the work in flight and wipes its staging. */
typedef struct mpe_gg20_pipeline mpe_gg20_pipeline;
int mpe_gg20_pipeline_create(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, int group, int lanes, int dedup_verify,
                             mpe_gg20_pipeline** out);
int mpe_gg20_pipeline_destroy(mpe_gg20_pipeline* p);
int mpe_gg20_pipeline_submit(mpe_gg20_pipeline* p, const int32_t* d_keyset, const mpe_gg20_nonces* nonces, uint32_t* d_r, uint32_t* d_s,
                             int32_t* d_recid, uint32_t* d_R, int32_t* d_status, void* stream, uint64_t* ticket);
int mpe_gg20_pipeline_submit_seeded(mpe_gg20_pipeline* p, const int32_t* d_keyset, const uint8_t* h_seed32, uint64_t batch_counter,
                                    const uint32_t* d_msg, uint32_t* d_r, uint32_t* d_s, int32_t* d_recid, uint32_t* d_R, int32_t* d_status,
                                    void* stream, uint64_t* ticket);
int mpe_gg20_pipeline_flush(mpe_gg20_pipeline* p);
int mpe_gg20_pipeline_query(mpe_gg20_pipeline* p, uint64_t ticket, int* done);
int mpe_gg20_pipeline_wait(mpe_gg20_pipeline* p, uint64_t ticket);
int mpe_gg20_pipeline_stream_wait(mpe_gg20_pipeline* p, uint64_t ticket, void* stream);
int mpe_gg20_pipeline_latency_ms(mpe_gg20_pipeline* p, uint64_t ticket, float* ms);
int mpe_gg20_pipeline_pass_ms(mpe_gg20_pipeline* p, uint64_t ticket, float* ms);
/* Failure contract.  A failing CHECK of a session is that session's status (above), never an error.  When the PASS that carries a
 * group fails as a whole (MPE_E_NOMEM / MPE_E_HIP from the sampler or the lock-step composition), every batch of that group is
 * told: its d_status array is filled with MPE_GG20_STATUS_PASS_FAILED(rc), its d_r / d_s / d_recid (/ d_R) are zeroed (on the
 * lane's stream, ordered before the ticket completes), and wait / stream_wait / latency_ms / pass_ms — and query once the ticket
 * is done — return that rc for EACH of its tickets.  submit itself returns MPE_OK when the batch was accepted: an error belongs
 * to the batches of the failed pass, not to the caller whose submission happened to close the group.  Later groups are unaffected.
 * (The reference reports per session and never drops an error: gg_2020/mod.rs:23-27, state_machine/sign/rounds.rs:696-713.)
 *   ticket_rc      *launched = 0 while the batch's group is still open; otherwise *rc = the pass's return code
 *   inject_fault   test hook: the next `passes` passes fail with rc (MPE_E_NOMEM or MPE_E_HIP) after their inputs were staged */
#define MPE_GG20_STATUS_PASS_FAILED(rc) (9000 - (rc))        /* 9001 MPE_E_ARG, 9002 MPE_E_HIP, 9003 MPE_E_NOMEM */
int mpe_gg20_pipeline_ticket_rc(mpe_gg20_pipeline* p, uint64_t ticket, int* launched, int* rc);
int mpe_gg20_pipeline_inject_fault(mpe_gg20_pipeline* p, int passes, int rc);
/* When a part-filled group goes to the device: when it is full, on flush / wait, and — evaluated inside every submit / query /
 * poll call of the ONE host thread that drives the pipeline, there is no hidden thread —
 *   set_deadline_us   when its oldest batch has waited `us` microseconds of host time (us < 0: never, the default);
 *   set_eager         as soon as the lane it would run on is idle: arrival-driven grouping — a trickle of batches starts at once,
 *                     under load the lanes are busy and the groups fill by themselves (off by default);
 *   poll              applies both rules now (a service calls it from its loop); *launched = 1 if a group went.
 * counters: groups launched in total / by the deadline / by an idle lane / failed passes. */
int mpe_gg20_pipeline_set_deadline_us(mpe_gg20_pipeline* p, int64_t us);
int mpe_gg20_pipeline_set_eager(mpe_gg20_pipeline* p, int on);
int mpe_gg20_pipeline_poll(mpe_gg20_pipeline* p, int* launched);
int mpe_gg20_pipeline_counters(const mpe_gg20_pipeline* p, uint64_t* groups, uint64_t* by_deadline, uint64_t* by_idle, uint64_t* failed);
/* items of the seeded submissions whose rejection loops gave up (see mpe_sample_below); waits for the lanes */
int mpe_gg20_pipeline_sampler_failures(mpe_gg20_pipeline* p, int32_t* h_out);


/* ---- multi-GPU: the per-round message fan-out behind the C-ABI (RCCL over xGMI; SURVEY.md 8e B, BASELINE config 5) ------------
 * Party-sharded signing: the parties of a session live on DIFFERENT GPUs (one process per GPU) and every round's messages travel
 * through ONE all-gather, after which each party reads what is addressed to it — the reference's relay broadcasts every message,
 * P2P ones included, to the room and the client filters (examples/gg20_sm_client.rs:35-40; state_machine/sign.rs:252-438).
 *   mpe_comm_unique_id   ncclGetUniqueId: ONE rank calls it, the host carries the MPE_COMM_ID_BYTES bytes to the others (any channel)
 *   mpe_comm_create      ncclCommInitRank on the context's device; blocks until all `world` ranks have called it
 *   mpe_comm_all_gather  d_buf = [world][bytes_per_rank]; this rank's block already sits at rank * bytes_per_rank; one ncclAllGather
 *                        queued on `stream` fills the others (in place; from a copy of the block when the self-test chose that form)
 *   mpe_comm_layout_self_test  one all-gather of known row patterns on the REAL communicator before any signing work: in place
 *                        first, then the copy form; all ranks agree (ncclAllReduce) on the first form that puts every row of every rank
 *                        where h_in_off will look for it.  *h_mode = 0 in place / 1 copy, *h_ok = 1; an error when neither is right.
 * Placement of party p (signer ordinal) of session block s:
 *   MPE_PLACE_PARTY    rank p % world, slot p / world (world divides S; one block);
 *   MPE_PLACE_ROTATED  world blocks; rank (s + p) % world, slot p — every rank hosts S (block, party) pairs for any world size, and
 *                      with world >= S no two parties of a session share a GPU.
 * The gather buffer is [world * per_rank] rows of [batch][W(round)] records; rank r owns rows [r * per_rank, (r + 1) * per_rank), row
 * r * per_rank + slot = that pair's outgoing records (mpe_gg20_roundN's d_out points there).  mpe_gg20_shard_in_off gives, for a
 * block, the record offset of every sender's row: the h_in_off argument of the next mpe_gg20_roundN, which reads the gathered
 * buffer in place.  mpe_gg20_round_exchange = the all-gather of round `round`'s records (per_rank * batch * W words per rank). */
#define MPE_COMM_ID_BYTES 128
#define MPE_PLACE_PARTY 0
#define MPE_PLACE_ROTATED 1
typedef struct mpe_comm mpe_comm;
int mpe_comm_unique_id(uint8_t* h_id);
int mpe_comm_create(mpe_ctx* ctx, const uint8_t* h_id, int rank, int world, mpe_comm** out);
int mpe_comm_destroy(mpe_comm* comm);
/* RCCL is bound at run time (dlopen) by the first communicator call — libmpecdsa_hip.so itself does not link it, so single-GPU
 * users need no librccl.  A copy the process ALREADY holds (PyTorch loads its own librccl.so.1) is adopted, so a process never
 * runs two RCCL instances; otherwise librccl.so.1 is searched on the loader path, then under $ROCM_PATH/lib (/opt/rocm/lib).
 * mpe_comm_library: the bound file (path_buf, NUL-terminated, may be NULL), *adopted = 1 if it was already loaded, *version =
 * ncclGetVersion.  MPE_E_HIP (and mpe_last_error) when no RCCL can be found. */
int mpe_comm_library(char* path_buf, size_t path_cap, int* adopted, int* version);
int mpe_comm_rank(const mpe_comm* comm);
int mpe_comm_world(const mpe_comm* comm);
int mpe_comm_gather_mode(const mpe_comm* comm);
int mpe_comm_all_gather(mpe_comm* comm, void* d_buf, size_t bytes_per_rank, void* stream);
int mpe_comm_layout_self_test(mpe_comm* comm, int rows_per_rank, int* h_mode, int* h_ok, void* stream);
int mpe_gg20_shard_where(int placement, int n_signers, int world, int block, int party, int* rank, int* slot);
int mpe_gg20_shard_blocks(int placement, int n_signers, int world);
int mpe_gg20_shard_per_rank(int placement, int n_signers, int world);
int mpe_gg20_shard_in_off(int placement, int n_signers, int world, int batch, int block, int64_t* h_in_off);
int mpe_gg20_round_exchange(mpe_comm* comm, int n_signers, int n, int round, int per_rank, int batch, uint32_t* d_slab, void* stream);

/* ---- keygen VERIFICATION math (src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:260-438) ----------------------- */
/* What every party checks about every other party's first keygen messages and shares, batched over (verifier, prover)
 * pairs / wallets.  Every key is its own modulus: the moduli set is built per call.  The two zk-paillier 0.4.3 proofs are
 * un-vendored; their definitions are recalled.
 * `NiCorrectKeyProof::verify(ek, SALT_STRING)` (party_i.rs:286-289): d_N [batch][64], d_sigma [batch][11][64]:
 *   sigma_i^N == rho_i (mod N) for the 11 hash-derived rho_i and no prime below 6370 divides N. */
int mpe_correct_key_verify(mpe_ctx* ctx, int batch, const uint32_t* d_N, const uint32_t* d_sigma, uint8_t* d_ok, void* stream);
/* `CompositeDLogProof{x, y}::verify(DLogStatement{N, g, ni})` (party_i.rs:294-301; called twice per prover, bases h1 and h2):
 *   N >= 2^128 and odd, gcd(g, N) = gcd(ni, N) = 1, e = H(x, g, N, ni), x == g^y ni^e mod N.  d_y [batch][73]. */
int mpe_composite_dlog_verify(mpe_ctx* ctx, int batch, const uint32_t* d_N, const uint32_t* d_g, const uint32_t* d_ni,
                              const uint32_t* d_x, const uint32_t* d_y, uint8_t* d_ok, void* stream);
/* The PROVE side of the same two proofs (what a party sends in keygen round 1, party_i.rs:219-258); key generation itself
 * (prime search) stays on the host.
 * `NiCorrectKeyProof::proof(dk, SALT_STRING)` for EVERY key of a private key set: d_sigma [nkeys][11][64],
 *   sigma_i = rho_i^(N^-1 mod phi(N)) mod N. */
int mpe_correct_key_prove(mpe_ctx* ctx, const mpe_paillier* sk, uint32_t* d_sigma, void* stream);
/* `CompositeDLogProof::prove(DLogStatement{N, g, ni}, secret)` with the sampled r (< 2^512, d_r [batch][16]) as input:
 *   x = g^r mod N, e = H(x, g, N, ni), y = r + e secret.  d_secret [batch][64]; outputs d_x [batch][64], d_y [batch][73]. */
int mpe_composite_dlog_prove(mpe_ctx* ctx, int batch, const uint32_t* d_N, const uint32_t* d_g, const uint32_t* d_ni,
                             const uint32_t* d_secret, const uint32_t* d_r, uint32_t* d_x, uint32_t* d_y, void* stream);
/* Feldman VSS (curv `VerifiableSS`): d_commits [batch][t1][16] (t1 = t + 1 coefficient commitments), d_index [batch]
 * (1-based party index).  validate_share (party_i.rs:337-340): share G == sum_k index^k C_k;
 * get_point_commitment (party_i.rs:383-385): that sum. */
int mpe_vss_validate_share(mpe_ctx* ctx, int batch, int t1, const uint32_t* d_commits, const uint32_t* d_share, const int32_t* d_index,
                           uint8_t* d_ok, void* stream);
int mpe_vss_point_commitment(mpe_ctx* ctx, int batch, int t1, const uint32_t* d_commits, const int32_t* d_index, uint32_t* d_out,
                             void* stream);
/* The two keygen verdicts AS THE REFERENCE COMPOSES THEM, batched over items = (keygen session, prover i): `n_parties` consecutive
 * items form one session, d_ok [batch], d_bad_actors [batch / n_parties] = the reference's `ErrorType::bad_actors` as a bit mask
 * over the provers of the session (may be NULL).
 * mpe_keygen_verify_round1 = `phase1_verify_com_phase3_verify_correct_key_verify_dlog_phase2_distribute` (party_i.rs:260-320):
 *   HashCommitment(y_i, blind_factor) == com  &&  NiCorrectKeyProof::verify(e)  &&  2047 <= e.n.bit_length() <= 2048  &&
 *   2047 <= dlog_statement.N.bit_length() <= 2048 (PAILLIER_MIN/MAX_BIT_LENGTH, party_i.rs:49-50; a wider modulus does not fit the
 *   64-word rows)  &&  composite_dlog_proof_base_h1.verify({N~, h1, h2})  &&  composite_dlog_proof_base_h2.verify({N~, h2, h1}).
 *   Rows: y [16], blind [8], com [8], N [64], sigma [11][64], Nt / h1 / h2 [64], x_h1 / x_h2 [64], y_h1 / y_h2 [73].
 * mpe_keygen_verify_round2 = the verdict of `phase2_verify_vss_construct_keypair_phase3_pok_dlog` (party_i.rs:322-367):
 *   vss_scheme_vec[i].validate_share(share_i, index)  &&  vss_scheme_vec[i].commitments[0] == y_vec[i].
 *   d_commits [batch][t1][16], d_share [batch][8], d_index [batch] (the VERIFIER's 1-based index), d_y [batch][16]. */
typedef struct {
  const uint32_t *y, *blind, *com, *N, *sigma, *Nt, *h1, *h2, *x_h1, *y_h1, *x_h2, *y_h2;
} mpe_keygen_round1;
int mpe_keygen_verify_round1(mpe_ctx* ctx, int batch, int n_parties, const mpe_keygen_round1* in, uint8_t* d_ok, uint32_t* d_bad_actors,
                             void* stream);
int mpe_keygen_verify_round2(mpe_ctx* ctx, int batch, int n_parties, int t1, const uint32_t* d_commits, const uint32_t* d_share,
                             const int32_t* d_index, const uint32_t* d_y, uint8_t* d_ok, uint32_t* d_bad_actors, void* stream);

/* ---- Lindell'17 two-party ECDSA, signing ---------------------------------------------- */
/* Party two, `PartialSig::compute(ek, encrypted_secret_share, local_share, ephemeral_local_share,
 * ephemeral_other_public_share, message)` (src/protocols/two_party_ecdsa/lindell_2017/party_two.rs:390-423).
 * pk = party ONE's Paillier key (public part is enough); c_key [batch][128] = Enc(x1); x2, k2, msg [batch][8];
 * R1 [batch][16] = party one's ephemeral public share; rho [batch][16] (< q^2, `BigInt::sample_below(&q.pow(2))`)
 * and r [batch][64] (the randomness `Paillier::encrypt` draws) are inputs.  c3 [batch][128]. */
int mpe_lindell_partial_sig(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx,
                            const uint32_t* d_c_key, const uint32_t* d_x2, const uint32_t* d_k2, const uint32_t* d_R1,
                            const uint32_t* d_msg, const uint32_t* d_rho, const uint32_t* d_r, uint32_t* d_c3, void* stream);
/* Party one, `Signature::compute_with_recid(party_one_private, partial_sig_c3, ephemeral_local_share,
 * ephemeral_other_public_share)` (lindell_2017/party_one.rs:519-565).  sk holds p, q; k1 [batch][8]; R2 [batch][16];
 * outputs r, s [batch][8] (s low: min(s, q - s)) and recid [batch]. */
int mpe_lindell_sign(mpe_ctx* ctx, const mpe_paillier* sk, int batch, const int32_t* d_key_idx, const uint32_t* d_c3,
                     const uint32_t* d_k1, const uint32_t* d_R2, uint32_t* d_r, uint32_t* d_s, int32_t* d_recid,
                     void* stream);

/* Key generation, the PDL exchange.  Party one, `pdl_proof(party1_private, paillier_key_pair)` (party_one.rs:366-401):
 * statement {ciphertext = c_key, ek, Q = x1 G, G = generator, h1, h2, N~}, witness {x1, c_key_randomness};
 * sk = party one's key set (private part), stm = its (N~, h1, h2) (create with mpe_statements_create); d_Q [batch][16] out.
 * (`CompositeDLogProof::prove` for (N~, h1, h2) is a host-side muladd + one mpe_modexp.) */
int mpe_lindell_pdl_proof(mpe_ctx* ctx, const mpe_paillier* sk, const mpe_statements* stm, int batch, const int32_t* d_key_idx,
                          const int32_t* d_st_idx, const uint32_t* d_c_key, const uint32_t* d_x1, const uint32_t* d_r,
                          const mpe_pdl_nonces* nonces, uint32_t* d_Q, const mpe_pdl_proof* out, void* stream);
/* Party two, `PaillierPublic::pdl_verify(composite_dlog_proof, statement, proof, paillier_public, q1)` (party_two.rs:275-300):
 * ok = statement.{ek, ciphertext, Q} == (pk[key], c_key, q1)  &&  CompositeDLogProof{x, y}.verify(DLogStatement{N~, h1, h2})
 *      &&  PDLwSlackProof::verify.  Every item brings its own statement: d_Nt, d_h1, d_h2, d_dlog_x [batch][64], d_dlog_y
 * [batch][73], d_stmt_N [batch][64], d_stmt_c / d_c_key [batch][128], d_stmt_Q / d_q1 [batch][16]. */
int mpe_lindell_pdl_verify(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx, const uint32_t* d_Nt,
                           const uint32_t* d_h1, const uint32_t* d_h2, const uint32_t* d_dlog_x, const uint32_t* d_dlog_y,
                           const uint32_t* d_stmt_N, const uint32_t* d_stmt_c, const uint32_t* d_stmt_Q, const uint32_t* d_c_key,
                           const uint32_t* d_q1, const mpe_pdl_proof* proof, uint8_t* d_ok, void* stream);

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
 * bench.py for the roofline line).  kind: 0 = modexp kernel (bits = modulus width), 1 = modmul kernel, 3 = N-adic pair kernel
 * (bits = width of the SQUARE: 4096 for N^2, 2048 for p^2 | q^2), 4 = the pair kernel in `half` mode (plain Montgomery ladder
 * modulo N or p; bits = that width), 5 = fixed-base ladder modulo N~ (exp2_words = the table's window width in bits),
 * 6 = the pair kernel with a PUBLIC exponent (the key N): items ordered by key, sliding windows wherever a wave shares it. */
typedef struct {
  int kind; int bits; int exp_words; int batch; float ms;
  int exp2_words;       /* != 0: mpe_modexp2-style launch (two bases on one ladder) */
  float sliding_frac;   /* kind 6: the share of the launch's (wave, trip) pairs that really ran the sliding-window schedule, counted by
                         * the kernel (waves that straddle a key boundary, and two-base ladders whose first window would dip below
                         * the second exponent, keep fixed windows); 0 for the other kinds; -1 = not counted */
} mpe_prof_rec;
int mpe_prof_enable(mpe_ctx* ctx, int on);       /* clears earlier records */
int mpe_prof_collect(mpe_ctx* ctx, mpe_prof_rec* out, int max_records, int* n_out);  /* waits for the events */

#ifdef __cplusplus
}
#endif
#endif /* MPECDSA_HIP_H */
