/* mpe_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement, over libgmp (the reference's own arithmetic engine: curv-kzen feature
 * `rust-gmp-kzen`, Cargo.toml:29,36), of the GG20 hot-path formulas of ZenGo-X/multi-party-ecdsa
 * v0.8.1.  Each function cites the reference file:line it follows.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may call this library; the product
 * (libmpecdsa_hip.so) never links or loads it.
 *
 * PARITY UNPINNED: the reference ships no golden vectors / KATs for this path (every test there
 * is a randomized round trip) and no Rust toolchain exists in this image, so
 * this oracle cannot be checked against the reference's own outputs.  It is pinned instead by
 * (1) an independent pure-Python restatement (tests/pyref.py: Python ints + hashlib),
 * (2) the reference's round-trip properties (prove->verify, MtA alpha+beta=ab, sign->verify),
 * (3) SHA-256 / secp256k1 published test vectors.
 *
 * Data convention = the product's C-ABI (include/mpecdsa_hip.h): little-endian uint32 words,
 * item-major batches.  All pointers are HOST pointers.  Randomness is always an explicit input
 * (the reference draws from OsRng inside the primitives; SURVEY.md §7 "Randomness").
 */
#ifndef MPE_ORACLE_H
#define MPE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* field widths in 32-bit words */
#define ORC_W256 8      /* scalars, e, SHA-256 digests            */
#define ORC_W768 24     /* alpha < q^3                             */
#define ORC_WS1 25      /* s1 = e*a + alpha < 2^769                */
#define ORC_W1024 32    /* Paillier primes p, q                    */
#define ORC_W2048 64    /* N, N~, h1, h2, z, s, plaintexts         */
#define ORC_W2304 72    /* rho < q*N~                              */
#define ORC_W2560 80    /* Bob's gamma < q^2*N                     */
#define ORC_WT1 81      /* Bob's t1 = e*beta' + gamma < 2^2561     */
#define ORC_W2816 88    /* gamma < q^3*N~                          */
#define ORC_WS2 89      /* s2 = e*rho + gamma < 2^2817             */
#define ORC_W4096 128   /* N^2, ciphertexts                        */
#define ORC_WPOINT 16   /* affine point: x[8] | y[8]; all-zero = point at infinity */

const char* orc_version(void);

/* The recalled byte-level conventions of curv-kzen 0.9 / zk-paillier 0.4.3 as a run-time profile — the oracle's mirror of
 * the product's mpe_encoding (include/mpecdsa_hip.h documents every field; same layout, so a test passes one ctypes struct to
 * both).  Process-wide; defaults = what this repository believes the crates do. */
typedef struct {
  uint8_t chain_point, zero_bytes, ck_mask_order, reserved;
  uint32_t ck_salt;
  uint8_t ord_dlog[4], ord_pedersen[8], ord_heg[8], ord_ecddh[8], ord_cdlog[4];
} orc_encoding;
void orc_set_encoding(const orc_encoding* e);
void orc_get_encoding(orc_encoding* out);

/* ---- curv BigInt (A.1) ---------------------------------------------------------------- */
/* out = base^exp mod m  (mpz_powm; BigInt::mod_pow).  mods: [nmods][k32]; mod_idx NULL -> (nmods==1?0:i) */
void orc_modexp(int k32, int batch, int nmods, const uint32_t* mods, const int32_t* mod_idx,
                const uint32_t* base, const uint32_t* exp, int exp_words, uint32_t* out);
void orc_modmul(int k32, int batch, int nmods, const uint32_t* mods, const int32_t* mod_idx,
                const uint32_t* a, const uint32_t* b, uint32_t* out);
/* out = a^-1 mod m, ok[i]=0 when not invertible (BigInt::mod_inv -> Option) */
void orc_modinv(int k32, int batch, int nmods, const uint32_t* mods, const int32_t* mod_idx,
                const uint32_t* a, uint32_t* out, uint8_t* ok);

/* ---- kzen-paillier (A.4) ---------------------------------------------------------------- */
/* c = (1 + m*N) * r^N mod N^2     Paillier::encrypt_with_chosen_randomness (mta/mod.rs:68-75,133-137) */
void orc_paillier_encrypt(int batch, int nkeys, const uint32_t* N /*[nkeys][64]*/, const int32_t* key_idx,
                          const uint32_t* m /*[B][64]*/, const uint32_t* r /*[B][64]*/, uint32_t* c /*[B][128]*/);
/* CRT decryption, Paillier::decrypt (mta/mod.rs:165) */
void orc_paillier_decrypt(int batch, int nkeys, const uint32_t* p /*[nkeys][32]*/, const uint32_t* q,
                          const int32_t* key_idx, const uint32_t* c /*[B][128]*/, uint32_t* m /*[B][64]*/);
/* c1*c2 mod N^2 (Paillier::add, mta/mod.rs:145);  c^k mod N^2 (Paillier::mul, mta/mod.rs:140-144; k: [B][64]) */
void orc_paillier_add(int batch, int nkeys, const uint32_t* N, const int32_t* key_idx,
                      const uint32_t* c1, const uint32_t* c2, uint32_t* out);
void orc_paillier_mul(int batch, int nkeys, const uint32_t* N, const int32_t* key_idx,
                      const uint32_t* c, const uint32_t* k, uint32_t* out);

/* ---- SHA-256 / curv DigestExt ------------------------------------------------------------ */
void orc_sha256(const uint8_t* msg, uint64_t len, uint8_t out[32]);

/* ---- secp256k1 (A.2) ---------------------------------------------------------------------- */
/* out = k*G (fixed base) ; out = k*P (variable base) ; scalars reduced mod q first (Scalar::from(&BigInt)) */
void orc_ec_mul_base(int batch, const uint32_t* k /*[B][8]*/, uint32_t* out /*[B][16]*/);
void orc_ec_mul(int batch, const uint32_t* k, const uint32_t* P /*[B][16]*/, uint32_t* out);
void orc_ec_add(int batch, const uint32_t* P, const uint32_t* Q, uint32_t* out);
void orc_ec_compress(int batch, const uint32_t* P, uint8_t* out33 /*[B][33]*/);

/* ---- MtA range proofs (src/utilities/mta/range_proofs.rs) --------------------------------- */
/* Statement tables: Nt/h1/h2 [nst][64] (DLogStatement{N,g,ni}, party_i.rs:225-229); ek N [nkeys][64]. */
/* AliceProof::generate  range_proofs.rs:160-193 (+ AliceZkpRound1/2 :39-90) */
void orc_alice_generate(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                        const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx,
                        const uint32_t* a /*[B][8]*/, const uint32_t* cipher /*[B][128]*/, const uint32_t* r /*[B][64]*/,
                        const uint32_t* alpha /*[B][24]*/, const uint32_t* beta /*[B][64]*/,
                        const uint32_t* gamma /*[B][88]*/, const uint32_t* rho /*[B][72]*/,
                        uint32_t* z /*[B][64]*/, uint32_t* e /*[B][8]*/, uint32_t* s /*[B][64]*/,
                        uint32_t* s1 /*[B][25]*/, uint32_t* s2 /*[B][89]*/);
/* AliceProof::verify  range_proofs.rs:105-156 */
void orc_alice_verify(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                      const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx, const uint32_t* cipher,
                      const uint32_t* z, const uint32_t* e, const uint32_t* s, const uint32_t* s1,
                      const uint32_t* s2, uint8_t* ok);

/* BobProof::generate range_proofs.rs:414-487 (+ BobZkpRound1/2 :218-297); check!=0 -> also u = alpha*G, X = b*G */
void orc_bob_generate(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                      const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx,
                      const uint32_t* a_enc /*[B][128]*/, const uint32_t* mta_enc /*[B][128]*/,
                      const uint32_t* b /*[B][8]*/, const uint32_t* beta_prim /*[B][64]*/, const uint32_t* r /*[B][64]*/,
                      const uint32_t* alpha /*[B][24]*/, const uint32_t* beta /*[B][64]*/, const uint32_t* gamma /*[B][80]*/,
                      const uint32_t* rho /*[B][72]*/, const uint32_t* rho_prim /*[B][88]*/,
                      const uint32_t* sigma /*[B][72]*/, const uint32_t* tau /*[B][88]*/, int check,
                      uint32_t* t /*[B][64]*/, uint32_t* z /*[B][64]*/, uint32_t* e /*[B][8]*/, uint32_t* s /*[B][64]*/,
                      uint32_t* s1 /*[B][25]*/, uint32_t* s2 /*[B][89]*/, uint32_t* t1 /*[B][81]*/, uint32_t* t2 /*[B][89]*/,
                      uint32_t* u /*[B][16] or NULL*/);
/* BobProof::verify range_proofs.rs:321-412; X,u non-NULL -> BobProofExt::verify :499-534 */
void orc_bob_verify(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                    const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx, const uint32_t* a_enc,
                    const uint32_t* mta_enc, const uint32_t* t, const uint32_t* z, const uint32_t* e,
                    const uint32_t* s, const uint32_t* s1, const uint32_t* s2, const uint32_t* t1,
                    const uint32_t* t2, const uint32_t* X, const uint32_t* u, uint8_t* ok);

/* ---- PDL with slack (src/utilities/zk_pdl_with_slack/mod.rs) ------------------------------ */
/* PDLwSlackProof::prove :68-125.  Q,G: statement points; x: witness scalar; r: Paillier randomness */
void orc_pdl_prove(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                   const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx,
                   const uint32_t* cipher, const uint32_t* Q, const uint32_t* G, const uint32_t* x, const uint32_t* r,
                   const uint32_t* alpha /*[B][24]*/, const uint32_t* beta /*[B][64]*/, const uint32_t* rho /*[B][72]*/,
                   const uint32_t* gamma /*[B][88]*/,
                   uint32_t* z /*[B][64]*/, uint32_t* u1 /*[B][16]*/, uint32_t* u2 /*[B][128]*/, uint32_t* u3 /*[B][64]*/,
                   uint32_t* s1 /*[B][25]*/, uint32_t* s2 /*[B][64]*/, uint32_t* s3 /*[B][89]*/);
/* PDLwSlackProof::verify :127-179 */
void orc_pdl_verify(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                    const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx, const uint32_t* cipher,
                    const uint32_t* Q, const uint32_t* G, const uint32_t* z, const uint32_t* u1, const uint32_t* u2,
                    const uint32_t* u3, const uint32_t* s1, const uint32_t* s2, const uint32_t* s3, uint8_t* ok);

/* ---- curv sigma proofs (A.3) --------------------------------------------------------------- */
/* DLogProof::prove(sk) with nonce rho: pk = sk*G, R = rho*G, c = H(R,G,pk), z = rho - c*sk */
void orc_dlog_prove(int batch, const uint32_t* sk, const uint32_t* nonce, uint32_t* pk, uint32_t* R, uint32_t* z);
void orc_dlog_verify(int batch, const uint32_t* pk, const uint32_t* R, const uint32_t* z, uint8_t* ok);

/* ---- curv sigma proofs used by GG20 phase 3 / phase 6 and the hash commitment (A.3), nonces as inputs ------ */
/* PedersenProof::prove(m, r): com = m G + r H (H = base_point2), a1 = s1 G, a2 = s2 H, e = H(G,H,com,a1,a2), z = s + e w */
void orc_pedersen_prove(int batch, const uint32_t* m, const uint32_t* r, const uint32_t* s1, const uint32_t* s2, uint32_t* com,
                        uint32_t* e, uint32_t* a1, uint32_t* a2, uint32_t* z1, uint32_t* z2);
void orc_pedersen_verify(int batch, const uint32_t* com, const uint32_t* a1, const uint32_t* a2, const uint32_t* z1,
                         const uint32_t* z2, uint8_t* ok);
/* HomoELGamalProof::prove(w{x,r}, delta{G,H,Y,D,E}): T = s1 H + s2 Y, A3 = s2 G, e = H(T,A3,G,H,Y,D,E) */
void orc_heg_prove(int batch, const uint32_t* x, const uint32_t* r, const uint32_t* s1, const uint32_t* s2, const uint32_t* G,
                   const uint32_t* H, const uint32_t* Y, const uint32_t* D, const uint32_t* E, uint32_t* T, uint32_t* A3,
                   uint32_t* z1, uint32_t* z2);
void orc_heg_verify(int batch, const uint32_t* G, const uint32_t* H, const uint32_t* Y, const uint32_t* D, const uint32_t* E,
                    const uint32_t* T, const uint32_t* A3, const uint32_t* z1, const uint32_t* z2, uint8_t* ok);
/* HashCommitment::create_commitment_with_user_defined_randomness(BigInt::from_bytes(P.to_bytes(true)), blind) -> [B][8] */
void orc_hash_commit_point(int batch, const uint32_t* P, const uint32_t* blind, uint32_t* com);

/* ---- GG20 signing, one party at a time (gg20_oracle.c): RoundN::proceed as functions of (state, messages) ------- */
/* keys: tables [nkeysets][n][..] (x: shares [8], p,q: Paillier primes [32], N: Paillier moduli [64] or NULL = p*q,
 * Nt,h1,h2 [64], X: pk_vec [16]) and y [nkeysets][16]; signers: S ascending party indices (t < S <= n).
 * A party object reads x, p, q of its own index only.
 * Nonces: leading dimensions [B][L] (L = parties whose nonces the arrays hold), then statement st (n), peer slot
 * jj (S-1; peer ordinal ind = jj < i ? jj : jj+1) and MessageB variant v (0 = gamma_i, 1 = w_i); msg [B][8]. */
typedef struct {
  int t, n, S, nkeysets;
  const int32_t* signers;
  const uint32_t *x, *p, *q, *N, *Nt, *h1, *h2, *y, *X;
} orc_gg20_keys;

typedef struct {
  const uint32_t *k, *gamma, *blind, *r_a;
  const uint32_t *al_alpha, *al_beta, *al_gamma, *al_rho;
  const uint32_t *mb_beta_tag, *mb_r, *mb_nonce_b, *mb_nonce_bt;
  const uint32_t *l, *ped_s1, *ped_s2;
  const uint32_t *pdl_alpha, *pdl_beta, *pdl_rho, *pdl_gamma;
  const uint32_t *heg_s1, *heg_s2;
  const uint32_t* msg;
} orc_gg20_nonces;

/* words of one (sender, session) record of the message Round `round` emits (0..5; 7 = PartialSignature) */
int orc_gg20_msg_words(int S, int n, int round);
typedef struct orc_gg20_party orc_gg20_party;
orc_gg20_party* orc_gg20_party_new(const orc_gg20_keys* K, int ord, int B, const orc_gg20_nonces* Z, int L, int li,
                                   const int32_t* keyset);
void orc_gg20_party_free(orc_gg20_party* P);
/* round 0..7 = RoundN::proceed / Round7::new, 8 = SignManual::complete, for sessions [first, first+count).
 * in: previous round's records of all S senders, sender j's [B][W] block at record offset in_off[j] (NULL: j*B);
 * out: this party's [B][W] block (NULL for rounds 6 and 8). */
void orc_gg20_party_round(orc_gg20_party* P, int round, const uint32_t* in, const int64_t* in_off, uint32_t* out, int first,
                          int count);
void orc_gg20_party_result(const orc_gg20_party* P, int32_t* status, uint32_t* bad_actors, uint32_t* r, uint32_t* s,
                           int32_t* recid, uint32_t* R);
void orc_gg20_party_fault(orc_gg20_party* P, int step);        /* the reference tests' corrupt_step 5 / 6 / 7 (0 = honest) */

/* all parties in lock-step (round_based::dev::Simulation).  slabs: NULL or 7 pointers (M0..M6) to [S][B][W];
 * party_status / party_bad: NULL or [S][B]; status[b] = smallest non-zero party status. */
void orc_gg20_sign_ex(const orc_gg20_keys* K, const orc_gg20_nonces* Z, const int32_t* keyset, int B, int first, int count,
                      uint32_t* const* slabs, uint32_t* r_out, uint32_t* s_out, int32_t* recid_out, uint32_t* R_out,
                      int32_t* status, int32_t* party_status, uint32_t* party_bad);
void orc_gg20_sign(const orc_gg20_keys* K, const orc_gg20_nonces* Z, int first, int count, uint32_t* r_out,
                   uint32_t* s_out, int32_t* recid_out, uint32_t* R_out, int32_t* status);

/* ---- identifiable abort (gg_2020/blame.rs), all-openings-in: bad[b] = bit mask over signer ordinals ---------------- */
void orc_ecddh_prove(int batch, const uint32_t* x, const uint32_t* s, const uint32_t* g1, const uint32_t* h1, const uint32_t* g2,
                     const uint32_t* h2, uint32_t* a1, uint32_t* a2, uint32_t* z);
void orc_ecddh_verify(int batch, const uint32_t* g1, const uint32_t* h1, const uint32_t* g2, const uint32_t* h2, const uint32_t* a1,
                      const uint32_t* a2, const uint32_t* z, uint8_t* ok);
/* Paillier::open: m [B][64], r [B][64] with c = (1 + m N) r^N mod N^2 */
void orc_paillier_open(int batch, int nkeys, const uint32_t* p, const uint32_t* q, const int32_t* key_idx, const uint32_t* c,
                       uint32_t* m, uint32_t* r);
typedef struct { const uint32_t *k, *k_rand, *gamma, *beta_tag, *beta_rand, *delta, *g_gamma, *c_a, *c_b; } orc_blame5_in;
typedef struct { const uint32_t *k, *k_rand, *miu, *miu_rand, *a1, *a2, *z, *S, *c_a, *c_b, *R; } orc_blame6_in;
typedef struct { const uint32_t *s, *r, *R_dash, *m, *R, *S; } orc_blame7_in;
void orc_gg20_blame5(const orc_gg20_keys* K, const int32_t* keyset, int B, const orc_blame5_in* in, uint32_t* bad);
void orc_gg20_blame6(const orc_gg20_keys* K, const int32_t* keyset, int B, const orc_blame6_in* in, uint32_t* bad);
void orc_gg20_blame7(int S, int B, const orc_blame7_in* in, uint32_t* bad);
void orc_gg20_party_sigma(const orc_gg20_party* P, uint32_t* sigma);

/* ---- keygen verification math (gg_2020/party_i.rs:260-438; zk-paillier proofs recalled, App. A.5) ------------------------ */
void orc_composite_dlog_verify(int batch, const uint32_t* N, const uint32_t* g, const uint32_t* ni, const uint32_t* x /*[B][64]*/,
                               const uint32_t* y /*[B][73]*/, uint8_t* ok);
void orc_composite_dlog_prove(int batch, const uint32_t* N, const uint32_t* g, const uint32_t* ni, const uint32_t* secret /*[B][64]*/,
                              const uint32_t* r /*[B][16]*/, uint32_t* x, uint32_t* y);
void orc_correct_key_verify(int batch, const uint32_t* N, const uint32_t* sigma /*[B][11][64]*/, uint8_t* ok);
void orc_correct_key_prove(int batch, const uint32_t* p, const uint32_t* q, uint32_t* sigma);
void orc_vss_validate_share(int batch, int t1, const uint32_t* commits /*[B][t1][16]*/, const uint32_t* share /*[B][8]*/,
                            const int32_t* index /*[B]*/, uint8_t* ok);
void orc_vss_point_commitment(int batch, int t1, const uint32_t* commits, const int32_t* index, uint32_t* out /*[B][16]*/);

/* the two keygen verdicts as the reference composes them (party_i.rs:260-320, 322-367); items = (session, prover), n_parties per session;
 * bad [batch / n_parties]: bit i = prover i is in `bad_actors` */
void orc_keygen_verify_round1(int batch, int n_parties, const uint32_t* y, const uint32_t* blind, const uint32_t* com, const uint32_t* N,
                              const uint32_t* sigma, const uint32_t* Nt, const uint32_t* h1, const uint32_t* h2, const uint32_t* x_h1,
                              const uint32_t* y_h1, const uint32_t* x_h2, const uint32_t* y_h2, uint8_t* ok, uint32_t* bad);
void orc_keygen_verify_round2(int batch, int n_parties, int t1, const uint32_t* commits, const uint32_t* share, const int32_t* index,
                              const uint32_t* y, uint8_t* ok, uint32_t* bad);

/* fixture helper (test key material only): smallest prime > start */
void orc_nextprime(int k32, const uint32_t* start, uint32_t* out);

/* Lindell'17 two-party signing (lindell_oracle.c): PartialSig::compute and Signature::compute_with_recid.
 * c_key, c3 [batch][128]; x2, k2, k1, msg [batch][8]; R1, R2 [batch][16]; rho [batch][16] (< q^2); r [batch][64]. */
void orc_lindell_partial_sig(int batch, int nkeys, const uint32_t* N, const int32_t* key_idx, const uint32_t* c_key,
                             const uint32_t* x2, const uint32_t* k2, const uint32_t* R1, const uint32_t* msg,
                             const uint32_t* rho, const uint32_t* r, uint32_t* c3);
void orc_lindell_sign(int batch, int nkeys, const uint32_t* p, const uint32_t* q, const int32_t* key_idx, const uint32_t* c3,
                      const uint32_t* k1, const uint32_t* R2, uint32_t* r_out, uint32_t* s_out, int32_t* recid);

#ifdef __cplusplus
}
#endif
/* ---- the device-side sampler's restatement (sampler_oracle.c): curv `Samplable`, `from_modulo`, `Scalar::random` over a ChaCha20
 * byte source; item i of stream sid reads key = seed, state[12] = block counter, state[13] = i, state[14..15] = sid ------------ */
void orc_chacha20_block(const uint8_t key[32], uint32_t counter, uint32_t n13, uint32_t n14, uint32_t n15, uint8_t out[64]);
int orc_sample_bits(int batch, const uint8_t* seed, uint64_t sid, int bits, int out_words, uint32_t* out);
/* rejection loops give up after n candidates (default 128), as the device sampler's option sampler_max_attempts */
void orc_sampler_set_max_attempts(int n);
/* BigInt::sample(bits) on given bytes: from_bytes_be(buf) >> (8 nbytes - bits) */
void orc_sample_rule(const uint8_t* buf, int nbytes, int bits, int out_words, uint32_t* out);
/* flags: 1 reject zero, 2 return 1 + draw (sample_range(1, bound + 1)), 4 from_modulo (repeat until gcd(x, bound) == 1); returns failures */
int orc_sample_below(int batch, const uint8_t* seed, uint64_t sid, const uint32_t* bound, int bound_words, int nbounds, const int32_t* bound_idx,
                     int flags, int out_words, uint32_t* out);
int orc_sample_scalar(int batch, const uint8_t* seed, uint64_t sid, uint32_t* out);
/* fills every field of Z but msg (the arrays are written: the const in orc_gg20_nonces is cast away) */
int orc_gg20_sample_nonces(const orc_gg20_keys* K, int B, int L, const int32_t* local, const int32_t* keyset, const uint8_t* seed, uint64_t counter,
                           const orc_gg20_nonces* Z);

#endif
