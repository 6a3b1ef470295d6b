/* mpe_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See mpe_oracle.h.
 * PARITY UNPINNED (no golden vectors exist in the reference; see header). */
#include "mpe_oracle.h"

#include <gmp.h>
#include <stdlib.h>
#include <string.h>

const char* orc_version(void) { return "mpe-oracle 0.1 (libgmp restatement; parity unpinned)"; }

/* ------------------------------------------------------------------------------------------ */
/* words <-> mpz                                                                               */
/* ------------------------------------------------------------------------------------------ */
static void zin(mpz_t x, const uint32_t* w, int n) { mpz_import(x, (size_t)n, -1, 4, 0, 0, w); }
static void zout(uint32_t* w, int n, const mpz_t x) {
  memset(w, 0, (size_t)n * 4);
  if (mpz_sizeinbase(x, 2) > (size_t)n * 32) abort();   /* oracle invariant: the field is wide enough */
  mpz_export(w, NULL, -1, 4, 0, 0, x);
}
static int pick(const int32_t* idx, int n, int i) { return idx ? idx[i] : (n == 1 ? 0 : i); }

/* ------------------------------------------------------------------------------------------ */
/* curv BigInt::mod_pow / mod_mul / mod_inv   (mpz_powm / mpz_invert)      */
/* ------------------------------------------------------------------------------------------ */
void orc_modexp(int k32, int batch, int nmods, const uint32_t* mods, const int32_t* mod_idx, const uint32_t* base,
                const uint32_t* exp, int exp_words, uint32_t* out) {
  mpz_t b, e, m, r;
  mpz_inits(b, e, m, r, NULL);
  for (int i = 0; i < batch; ++i) {
    zin(m, mods + (size_t)pick(mod_idx, nmods, i) * k32, k32);
    zin(b, base + (size_t)i * k32, k32);
    zin(e, exp + (size_t)i * exp_words, exp_words);
    mpz_powm(r, b, e, m);
    zout(out + (size_t)i * k32, k32, r);
  }
  mpz_clears(b, e, m, r, NULL);
}

void orc_modmul(int k32, int batch, int nmods, const uint32_t* mods, const int32_t* mod_idx, const uint32_t* a,
                const uint32_t* b, uint32_t* out) {
  mpz_t x, y, m;
  mpz_inits(x, y, m, NULL);
  for (int i = 0; i < batch; ++i) {
    zin(m, mods + (size_t)pick(mod_idx, nmods, i) * k32, k32);
    zin(x, a + (size_t)i * k32, k32);
    zin(y, b + (size_t)i * k32, k32);
    mpz_mul(x, x, y);
    mpz_mod(x, x, m);
    zout(out + (size_t)i * k32, k32, x);
  }
  mpz_clears(x, y, m, NULL);
}

void orc_modinv(int k32, int batch, int nmods, const uint32_t* mods, const int32_t* mod_idx, const uint32_t* a,
                uint32_t* out, uint8_t* ok) {
  mpz_t x, m;
  mpz_inits(x, m, NULL);
  for (int i = 0; i < batch; ++i) {
    zin(m, mods + (size_t)pick(mod_idx, nmods, i) * k32, k32);
    zin(x, a + (size_t)i * k32, k32);
    ok[i] = (uint8_t)(mpz_invert(x, x, m) != 0);
    if (!ok[i]) mpz_set_ui(x, 0);
    zout(out + (size_t)i * k32, k32, x);
  }
  mpz_clears(x, m, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* kzen-paillier 0.4.2                                                    */
/* ------------------------------------------------------------------------------------------ */
/* c = (1 + m*n) * r^n mod n^2 */
static void paillier_enc(mpz_t c, const mpz_t n, const mpz_t nn, const mpz_t m, const mpz_t r) {
  mpz_t rn, gm;
  mpz_inits(rn, gm, NULL);
  mpz_powm(rn, r, n, nn);
  mpz_mul(gm, m, n);
  mpz_add_ui(gm, gm, 1);
  mpz_mod(gm, gm, nn);
  mpz_mul(c, gm, rn);
  mpz_mod(c, c, nn);
  mpz_clears(rn, gm, NULL);
}
void orc_paillier_encrypt(int batch, int nkeys, const uint32_t* N, const int32_t* key_idx, const uint32_t* m,
                          const uint32_t* r, uint32_t* c) {
  mpz_t n, nn, mm, rr, cc;
  mpz_inits(n, nn, mm, rr, cc, NULL);
  for (int i = 0; i < batch; ++i) {
    zin(n, N + (size_t)pick(key_idx, nkeys, i) * ORC_W2048, ORC_W2048);
    mpz_mul(nn, n, n);
    zin(mm, m + (size_t)i * ORC_W2048, ORC_W2048);
    zin(rr, r + (size_t)i * ORC_W2048, ORC_W2048);
    paillier_enc(cc, n, nn, mm, rr);
    zout(c + (size_t)i * ORC_W4096, ORC_W4096, cc);
  }
  mpz_clears(n, nn, mm, rr, cc, NULL);
}

/* kzen-paillier `h(p, pp, n)`: L_p((1 - n) mod p^2)^-1 mod p, with L(u) = (u-1)/p */
static void paillier_h(mpz_t h, const mpz_t p, const mpz_t pp, const mpz_t n) {
  mpz_t g;
  mpz_init(g);
  mpz_ui_sub(g, 1, n);
  mpz_mod(g, g, pp);
  mpz_sub_ui(g, g, 1);
  mpz_divexact(g, g, p);
  mpz_invert(h, g, p);
  mpz_clear(g);
}
static void paillier_dec(mpz_t m, const mpz_t p, const mpz_t q, const mpz_t c) {
  mpz_t pp, qq, n, pinv, hp, hq, cp, cq, mp, mq, t;
  mpz_inits(pp, qq, n, pinv, hp, hq, cp, cq, mp, mq, t, NULL);
  mpz_mul(pp, p, p);
  mpz_mul(qq, q, q);
  mpz_mul(n, p, q);
  mpz_invert(pinv, p, q);
  paillier_h(hp, p, pp, n);
  paillier_h(hq, q, qq, n);
  mpz_mod(cp, c, pp);                       /* crt_decompose */
  mpz_mod(cq, c, qq);
  mpz_sub_ui(t, p, 1);
  mpz_powm(mp, cp, t, pp);
  mpz_sub_ui(mp, mp, 1);
  mpz_divexact(mp, mp, p);                  /* L_p */
  mpz_mul(mp, mp, hp);
  mpz_mod(mp, mp, p);
  mpz_sub_ui(t, q, 1);
  mpz_powm(mq, cq, t, qq);
  mpz_sub_ui(mq, mq, 1);
  mpz_divexact(mq, mq, q);
  mpz_mul(mq, mq, hq);
  mpz_mod(mq, mq, q);
  mpz_sub(t, mq, mp);                       /* crt_recombine */
  mpz_mod(t, t, q);
  mpz_mul(t, t, pinv);
  mpz_mod(t, t, q);
  mpz_mul(t, t, p);
  mpz_add(m, mp, t);
  mpz_clears(pp, qq, n, pinv, hp, hq, cp, cq, mp, mq, t, NULL);
}
void orc_paillier_decrypt(int batch, int nkeys, const uint32_t* p, const uint32_t* q, const int32_t* key_idx,
                          const uint32_t* c, uint32_t* m) {
  mpz_t pp, qq, cc, mm;
  mpz_inits(pp, qq, cc, mm, NULL);
  for (int i = 0; i < batch; ++i) {
    const int k = pick(key_idx, nkeys, i);
    zin(pp, p + (size_t)k * ORC_W1024, ORC_W1024);
    zin(qq, q + (size_t)k * ORC_W1024, ORC_W1024);
    zin(cc, c + (size_t)i * ORC_W4096, ORC_W4096);
    paillier_dec(mm, pp, qq, cc);
    zout(m + (size_t)i * ORC_W2048, ORC_W2048, mm);
  }
  mpz_clears(pp, qq, cc, mm, NULL);
}
void orc_paillier_add(int batch, int nkeys, const uint32_t* N, const int32_t* key_idx, const uint32_t* c1,
                      const uint32_t* c2, uint32_t* out) {
  mpz_t n, nn, a, b;
  mpz_inits(n, nn, a, b, NULL);
  for (int i = 0; i < batch; ++i) {
    zin(n, N + (size_t)pick(key_idx, nkeys, i) * ORC_W2048, ORC_W2048);
    mpz_mul(nn, n, n);
    zin(a, c1 + (size_t)i * ORC_W4096, ORC_W4096);
    zin(b, c2 + (size_t)i * ORC_W4096, ORC_W4096);
    mpz_mul(a, a, b);
    mpz_mod(a, a, nn);
    zout(out + (size_t)i * ORC_W4096, ORC_W4096, a);
  }
  mpz_clears(n, nn, a, b, NULL);
}
void orc_paillier_mul(int batch, int nkeys, const uint32_t* N, const int32_t* key_idx, const uint32_t* c,
                      const uint32_t* k, uint32_t* out) {
  mpz_t n, nn, a, e;
  mpz_inits(n, nn, a, e, NULL);
  for (int i = 0; i < batch; ++i) {
    zin(n, N + (size_t)pick(key_idx, nkeys, i) * ORC_W2048, ORC_W2048);
    mpz_mul(nn, n, n);
    zin(a, c + (size_t)i * ORC_W4096, ORC_W4096);
    zin(e, k + (size_t)i * ORC_W2048, ORC_W2048);
    mpz_powm(a, a, e, nn);
    zout(out + (size_t)i * ORC_W4096, ORC_W4096, a);
  }
  mpz_clears(n, nn, a, e, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* SHA-256 (FIPS 180-4) and curv DigestExt (SURVEY.md §8b "Hash", App. A.1)                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; } sha_t;
static const uint32_t SK[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,
  0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,
  0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,
  0x06ca6351,0x14292967,0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
  0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,0x19a4c116,0x1e376c08,
  0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,
  0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(sha_t* s, const uint8_t* p) {
  uint32_t w[64], a, b, c, d, e, f, g, h;
  for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4*i] << 24 | (uint32_t)p[4*i+1] << 16 | (uint32_t)p[4*i+2] << 8 | p[4*i+3];
  for (int i = 16; i < 64; ++i) {
    uint32_t s0 = ROR(w[i-15], 7) ^ ROR(w[i-15], 18) ^ (w[i-15] >> 3);
    uint32_t s1 = ROR(w[i-2], 17) ^ ROR(w[i-2], 19) ^ (w[i-2] >> 10);
    w[i] = w[i-16] + s0 + w[i-7] + s1;
  }
  a = s->h[0]; b = s->h[1]; c = s->h[2]; d = s->h[3]; e = s->h[4]; f = s->h[5]; g = s->h[6]; h = s->h[7];
  for (int i = 0; i < 64; ++i) {
    uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25), ch = (e & f) ^ (~e & g);
    uint32_t t1 = h + S1 + ch + SK[i] + w[i];
    uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s->h[0] += a; s->h[1] += b; s->h[2] += c; s->h[3] += d; s->h[4] += e; s->h[5] += f; s->h[6] += g; s->h[7] += h;
}
static void sha_init(sha_t* s) {
  static const uint32_t iv[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
  memcpy(s->h, iv, sizeof iv);
  s->len = 0;
}
static void sha_update(sha_t* s, const uint8_t* p, size_t n) {
  while (n) {
    size_t off = s->len & 63, take = 64 - off;
    if (take > n) take = n;
    memcpy(s->buf + off, p, take);
    s->len += take; p += take; n -= take;
    if ((s->len & 63) == 0) sha_block(s, s->buf);
  }
}
static void sha_final(sha_t* s, uint8_t out[32]) {
  uint64_t bits = s->len * 8;
  uint8_t pad = 0x80, z = 0, lenb[8];
  sha_update(s, &pad, 1);
  while ((s->len & 63) != 56) sha_update(s, &z, 1);
  for (int i = 0; i < 8; ++i) lenb[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha_update(s, lenb, 8);
  for (int i = 0; i < 8; ++i) { out[4*i] = s->h[i] >> 24; out[4*i+1] = s->h[i] >> 16; out[4*i+2] = s->h[i] >> 8; out[4*i+3] = s->h[i]; }
}
void orc_sha256(const uint8_t* msg, uint64_t len, uint8_t out[32]) {
  sha_t s; sha_init(&s); sha_update(&s, msg, (size_t)len); sha_final(&s, out);
}
/* The byte-level conventions of the un-vendored crates that the reference's source does not fix (the same struct, field for
 * field, as mpe_encoding of include/mpecdsa_hip.h — kept textually separate: the oracle never includes product headers).
 * Process-wide: set once by the test harness before the threads of a batch start. */
static orc_encoding ORC_ENC = {0, 0, 0, 0, 0x4B5A656Eu, {0, 1, 2, 3}, {0, 1, 2, 3, 4, 5, 6, 7}, {0, 1, 2, 3, 4, 5, 6, 7},
                               {0, 1, 2, 3, 4, 5, 6, 7}, {0, 1, 2, 3}};
void orc_set_encoding(const orc_encoding* e) { if (e) ORC_ENC = *e; }
void orc_get_encoding(orc_encoding* out) { if (out) *out = ORC_ENC; }

/* DigestExt::chain_bigint: update(BigInt::to_bytes()) = big-endian magnitude, minimal length
 * (rust-gmp exports 0 as one 0x00 byte — or, ORC_ENC.zero_bytes = 1, as nothing). */
static void chain_bigint(sha_t* s, const mpz_t x) {
  uint8_t buf[1024];
  size_t cnt = (mpz_sizeinbase(x, 2) + 7) / 8;
  if (mpz_sgn(x) == 0 && ORC_ENC.zero_bytes) cnt = 0;
  if (cnt > sizeof buf) abort();
  memset(buf, 0, cnt);
  mpz_export(buf, NULL, 1, 1, 0, 0, x);
  sha_update(s, buf, cnt);
}
/* DigestExt::result_bigint: digest as big-endian integer */
static void result_bigint(sha_t* s, mpz_t out) {
  uint8_t d[32];
  sha_final(s, d);
  mpz_import(out, 32, 1, 1, 0, 0, d);
}

/* ------------------------------------------------------------------------------------------ */
/* secp256k1 over mpz, affine.  Point at infinity: inf = 1.               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { mpz_t x, y; int inf; } pt_t;
static mpz_t EC_P, EC_Q, EC_GX, EC_GY;
static int ec_ready = 0;
static void ec_setup(void) {
  if (ec_ready) return;
  mpz_init_set_str(EC_P, "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F", 16);
  mpz_init_set_str(EC_Q, "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141", 16);
  mpz_init_set_str(EC_GX, "79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798", 16);
  mpz_init_set_str(EC_GY, "483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8", 16);
  ec_ready = 1;
}
static void pt_init(pt_t* p) { mpz_inits(p->x, p->y, NULL); p->inf = 1; }
static void pt_clear(pt_t* p) { mpz_clears(p->x, p->y, NULL); }
static void pt_set(pt_t* r, const pt_t* a) { mpz_set(r->x, a->x); mpz_set(r->y, a->y); r->inf = a->inf; }
static void pt_gen(pt_t* r) { ec_setup(); mpz_set(r->x, EC_GX); mpz_set(r->y, EC_GY); r->inf = 0; }
static void pt_add(pt_t* r, const pt_t* a, const pt_t* b) {
  ec_setup();
  if (a->inf) { pt_set(r, b); return; }
  if (b->inf) { pt_set(r, a); return; }
  mpz_t l, t, x3, y3;
  mpz_inits(l, t, x3, y3, NULL);
  if (mpz_cmp(a->x, b->x) == 0) {
    mpz_add(t, a->y, b->y);
    mpz_mod(t, t, EC_P);
    if (mpz_sgn(t) == 0) { r->inf = 1; mpz_set_ui(r->x, 0); mpz_set_ui(r->y, 0); goto done; }
    mpz_mul(l, a->x, a->x); mpz_mul_ui(l, l, 3);          /* 3x^2 / 2y */
    mpz_mul_ui(t, a->y, 2); mpz_invert(t, t, EC_P);
  } else {
    mpz_sub(l, b->y, a->y);
    mpz_sub(t, b->x, a->x); mpz_mod(t, t, EC_P); mpz_invert(t, t, EC_P);
  }
  mpz_mul(l, l, t); mpz_mod(l, l, EC_P);
  mpz_mul(x3, l, l); mpz_sub(x3, x3, a->x); mpz_sub(x3, x3, b->x); mpz_mod(x3, x3, EC_P);
  mpz_sub(y3, a->x, x3); mpz_mul(y3, y3, l); mpz_sub(y3, y3, a->y); mpz_mod(y3, y3, EC_P);
  mpz_set(r->x, x3); mpz_set(r->y, y3); r->inf = 0;
done:
  mpz_clears(l, t, x3, y3, NULL);
}
static void pt_neg(pt_t* r, const pt_t* a) {
  ec_setup();
  pt_set(r, a);
  if (!r->inf && mpz_sgn(r->y) != 0) mpz_sub(r->y, EC_P, r->y);
}
/* Jacobian helpers for the scalar multiplication (a = 0 curve): the result is converted back to affine, so the
 * points every caller sees are the unique affine ones whatever the ladder */
typedef struct { mpz_t X, Y, Z; int inf; } jac_t;
static void jac_init(jac_t* j) { mpz_inits(j->X, j->Y, j->Z, NULL); j->inf = 1; }
static void jac_clear(jac_t* j) { mpz_clears(j->X, j->Y, j->Z, NULL); }
static void jac_dbl(jac_t* r, const jac_t* a, mpz_t* t) {           /* t: 4 scratch values */
  if (a->inf || mpz_sgn(a->Y) == 0) { r->inf = 1; return; }
  mpz_mul(t[0], a->Y, a->Y); mpz_mod(t[0], t[0], EC_P);              /* Y^2 */
  mpz_mul(t[1], a->X, t[0]); mpz_mul_ui(t[1], t[1], 4); mpz_mod(t[1], t[1], EC_P);   /* S = 4 X Y^2 */
  mpz_mul(t[2], a->X, a->X); mpz_mul_ui(t[2], t[2], 3); mpz_mod(t[2], t[2], EC_P);   /* M = 3 X^2 */
  mpz_mul(t[3], a->Y, a->Z); mpz_mul_ui(t[3], t[3], 2); mpz_mod(t[3], t[3], EC_P);   /* Z' = 2 Y Z */
  mpz_mul(r->X, t[2], t[2]); mpz_submul_ui(r->X, t[1], 2); mpz_mod(r->X, r->X, EC_P);
  mpz_mul(t[0], t[0], t[0]); mpz_mul_ui(t[0], t[0], 8);             /* 8 Y^4 */
  mpz_sub(t[1], t[1], r->X); mpz_mul(t[1], t[1], t[2]); mpz_sub(t[1], t[1], t[0]); mpz_mod(r->Y, t[1], EC_P);
  mpz_set(r->Z, t[3]); r->inf = 0;
}
/* r = a + (x, y) affine; r may alias a */
static void jac_add_aff(jac_t* r, const jac_t* a, const pt_t* b, mpz_t* t) {       /* t: 6 scratch values */
  if (b->inf) { if (r != a) { mpz_set(r->X, a->X); mpz_set(r->Y, a->Y); mpz_set(r->Z, a->Z); r->inf = a->inf; } return; }
  if (a->inf) { mpz_set(r->X, b->x); mpz_set(r->Y, b->y); mpz_set_ui(r->Z, 1); r->inf = 0; return; }
  mpz_mul(t[0], a->Z, a->Z); mpz_mod(t[0], t[0], EC_P);                              /* Z^2 */
  mpz_mul(t[1], b->x, t[0]); mpz_mod(t[1], t[1], EC_P);                              /* U2 */
  mpz_mul(t[2], t[0], a->Z); mpz_mul(t[2], t[2], b->y); mpz_mod(t[2], t[2], EC_P);   /* S2 */
  mpz_sub(t[1], t[1], a->X); mpz_mod(t[1], t[1], EC_P);                              /* H */
  mpz_sub(t[2], t[2], a->Y); mpz_mod(t[2], t[2], EC_P);                              /* r */
  if (mpz_sgn(t[1]) == 0) {
    if (mpz_sgn(t[2]) == 0) { jac_t c; jac_init(&c); mpz_set(c.X, a->X); mpz_set(c.Y, a->Y); mpz_set(c.Z, a->Z); c.inf = 0; jac_dbl(r, &c, t); jac_clear(&c); }
    else r->inf = 1;
    return;
  }
  mpz_mul(t[3], t[1], t[1]); mpz_mod(t[3], t[3], EC_P);                              /* H^2 */
  mpz_mul(t[4], t[3], t[1]); mpz_mod(t[4], t[4], EC_P);                              /* H^3 */
  mpz_mul(t[3], t[3], a->X); mpz_mod(t[3], t[3], EC_P);                              /* X1 H^2 */
  mpz_mul(t[5], a->Z, t[1]); mpz_mod(t[5], t[5], EC_P);                              /* Z3 */
  mpz_mul(t[0], t[2], t[2]); mpz_sub(t[0], t[0], t[4]); mpz_submul_ui(t[0], t[3], 2); mpz_mod(t[0], t[0], EC_P);   /* X3 */
  mpz_sub(t[3], t[3], t[0]); mpz_mul(t[3], t[3], t[2]); mpz_mul(t[4], t[4], a->Y); mpz_sub(t[3], t[3], t[4]); mpz_mod(t[3], t[3], EC_P);
  mpz_set(r->X, t[0]); mpz_set(r->Y, t[3]); mpz_set(r->Z, t[5]); r->inf = 0;
}
/* r = k*P with k reduced mod q (Scalar::from(&BigInt) reduces; Point * Scalar): 4-bit windows over affine multiples */
static void pt_mul(pt_t* r, const mpz_t k, const pt_t* p) {
  ec_setup();
  mpz_t kk, t[6]; mpz_init(kk); mpz_mod(kk, k, EC_Q);
  for (int i = 0; i < 6; ++i) mpz_init(t[i]);
  if (p->inf || mpz_sgn(kk) == 0) { r->inf = 1; mpz_set_ui(r->x, 0); mpz_set_ui(r->y, 0); goto out; }
  {
    pt_t tab[16];
    for (int i = 0; i < 16; ++i) pt_init(&tab[i]);
    pt_set(&tab[1], p);
    for (int i = 2; i < 16; ++i) pt_add(&tab[i], &tab[i - 1], p);
    jac_t acc, tmp; jac_init(&acc); jac_init(&tmp);
    const int nb = (int)mpz_sizeinbase(kk, 2);
    for (int w = (nb + 3) / 4 - 1; w >= 0; --w) {
      for (int d = 0; d < 4; ++d) { jac_dbl(&tmp, &acc, t); mpz_swap(acc.X, tmp.X); mpz_swap(acc.Y, tmp.Y); mpz_swap(acc.Z, tmp.Z); acc.inf = tmp.inf; }
      const int dig = (int)(mpz_tstbit(kk, 4 * w) | mpz_tstbit(kk, 4 * w + 1) << 1 | mpz_tstbit(kk, 4 * w + 2) << 2 | mpz_tstbit(kk, 4 * w + 3) << 3);
      if (dig) jac_add_aff(&acc, &acc, &tab[dig], t);
    }
    if (acc.inf) { r->inf = 1; mpz_set_ui(r->x, 0); mpz_set_ui(r->y, 0); }
    else {
      mpz_invert(t[0], acc.Z, EC_P);
      mpz_mul(t[1], t[0], t[0]); mpz_mod(t[1], t[1], EC_P);
      mpz_mul(t[2], acc.X, t[1]); mpz_mod(t[2], t[2], EC_P);
      mpz_mul(t[1], t[1], t[0]); mpz_mul(t[1], t[1], acc.Y); mpz_mod(t[1], t[1], EC_P);
      mpz_set(r->x, t[2]); mpz_set(r->y, t[1]); r->inf = 0;
    }
    jac_clear(&acc); jac_clear(&tmp);
    for (int i = 0; i < 16; ++i) pt_clear(&tab[i]);
  }
out:
  for (int i = 0; i < 6; ++i) mpz_clear(t[i]);
  mpz_clear(kk);
}
static int pt_eq(const pt_t* a, const pt_t* b) {
  if (a->inf || b->inf) return a->inf && b->inf;
  return mpz_cmp(a->x, b->x) == 0 && mpz_cmp(a->y, b->y) == 0;
}
static void pt_in(pt_t* p, const uint32_t* w) {
  zin(p->x, w, 8); zin(p->y, w + 8, 8);
  p->inf = (mpz_sgn(p->x) == 0 && mpz_sgn(p->y) == 0);
}
static void pt_out(uint32_t* w, const pt_t* p) {
  if (p->inf) { memset(w, 0, 64); return; }
  zout(w, 8, p->x); zout(w + 8, 8, p->y);
}
/* Point::to_bytes(true): 33-byte SEC1 compressed; to_bytes(false): 65-byte uncompressed */
static void pt_bytes(const pt_t* p, int compressed, uint8_t* out) {
  uint8_t xb[32] = {0}, yb[32] = {0};
  size_t nx = (mpz_sizeinbase(p->x, 2) + 7) / 8, ny = (mpz_sizeinbase(p->y, 2) + 7) / 8;
  if (mpz_sgn(p->x)) mpz_export(xb + 32 - nx, NULL, 1, 1, 0, 0, p->x);
  if (mpz_sgn(p->y)) mpz_export(yb + 32 - ny, NULL, 1, 1, 0, 0, p->y);
  if (compressed) { out[0] = mpz_tstbit(p->y, 0) ? 3 : 2; memcpy(out + 1, xb, 32); }
  else { out[0] = 4; memcpy(out + 1, xb, 32); memcpy(out + 33, yb, 32); }
}
/* BigInt::from_bytes(P.to_bytes(true)) — how zk_pdl_with_slack hashes points (mod.rs:102-110) */
static void pt_as_bigint(mpz_t out, const pt_t* p) {
  uint8_t b[33]; pt_bytes(p, 1, b); mpz_import(out, 33, 1, 1, 0, 0, b);
}
/* DigestExt::chain_point: update(P.to_bytes(false))  [RECALLED, App. A.2; ORC_ENC.chain_point = 1: to_bytes(true)] */
static void chain_point(sha_t* s, const pt_t* p) {
  uint8_t b[65];
  pt_bytes(p, ORC_ENC.chain_point, b);
  sha_update(s, b, ORC_ENC.chain_point ? 33 : 65);
}

void orc_ec_mul_base(int batch, const uint32_t* k, uint32_t* out) {
  mpz_t kk; mpz_init(kk); pt_t g, r; pt_init(&g); pt_init(&r); pt_gen(&g);
  for (int i = 0; i < batch; ++i) { zin(kk, k + (size_t)i * 8, 8); pt_mul(&r, kk, &g); pt_out(out + (size_t)i * 16, &r); }
  pt_clear(&g); pt_clear(&r); mpz_clear(kk);
}
void orc_ec_mul(int batch, const uint32_t* k, const uint32_t* P, uint32_t* out) {
  mpz_t kk; mpz_init(kk); pt_t p, r; pt_init(&p); pt_init(&r);
  for (int i = 0; i < batch; ++i) {
    zin(kk, k + (size_t)i * 8, 8); pt_in(&p, P + (size_t)i * 16); pt_mul(&r, kk, &p); pt_out(out + (size_t)i * 16, &r);
  }
  pt_clear(&p); pt_clear(&r); mpz_clear(kk);
}
void orc_ec_add(int batch, const uint32_t* P, const uint32_t* Q, uint32_t* out) {
  pt_t p, q, r; pt_init(&p); pt_init(&q); pt_init(&r);
  for (int i = 0; i < batch; ++i) {
    pt_in(&p, P + (size_t)i * 16); pt_in(&q, Q + (size_t)i * 16); pt_add(&r, &p, &q); pt_out(out + (size_t)i * 16, &r);
  }
  pt_clear(&p); pt_clear(&q); pt_clear(&r);
}
void orc_ec_compress(int batch, const uint32_t* P, uint8_t* out33) {
  pt_t p; pt_init(&p);
  for (int i = 0; i < batch; ++i) { pt_in(&p, P + (size_t)i * 16); pt_bytes(&p, 1, out33 + (size_t)i * 33); }
  pt_clear(&p);
}

/* ------------------------------------------------------------------------------------------ */
/* statement / key helpers                                                                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct { mpz_t N, NN, Nt, h1, h2, q, q3; } env_t;
static void env_init(env_t* e) { mpz_inits(e->N, e->NN, e->Nt, e->h1, e->h2, e->q, e->q3, NULL); ec_setup(); mpz_set(e->q, EC_Q); mpz_pow_ui(e->q3, EC_Q, 3); }
static void env_clear(env_t* e) { mpz_clears(e->N, e->NN, e->Nt, e->h1, e->h2, e->q, e->q3, NULL); }
static void env_load(env_t* e, int i, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                     const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx) {
  const int k = pick(key_idx, nkeys, i), s = pick(st_idx, nst, i);
  zin(e->N, N + (size_t)k * ORC_W2048, ORC_W2048);
  mpz_mul(e->NN, e->N, e->N);
  zin(e->Nt, Nt + (size_t)s * ORC_W2048, ORC_W2048);
  zin(e->h1, h1 + (size_t)s * ORC_W2048, ORC_W2048);
  zin(e->h2, h2 + (size_t)s * ORC_W2048, ORC_W2048);
}
/* commitment_unknown_order(h1,h2,M,x,r) = h1^x * h2^r mod M, negative r via inverse
 * (zk_pdl_with_slack/mod.rs:182-199) */
static void commit_unknown_order(mpz_t out, const mpz_t h1, const mpz_t h2, const mpz_t M, const mpz_t x, const mpz_t r) {
  mpz_t a, b, t;
  mpz_inits(a, b, t, NULL);
  mpz_powm(a, h1, x, M);
  if (mpz_sgn(r) < 0) {
    mpz_invert(t, h2, M);
    mpz_neg(b, r);
    mpz_powm(b, t, b, M);
  } else {
    mpz_powm(b, h2, r, M);
  }
  mpz_mul(a, a, b);
  mpz_mod(out, a, M);
  mpz_clears(a, b, t, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* AliceProof (src/utilities/mta/range_proofs.rs:39-193)                                       */
/* ------------------------------------------------------------------------------------------ */
void orc_alice_generate(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                        const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx, const uint32_t* a,
                        const uint32_t* cipher, const uint32_t* r, const uint32_t* alpha, const uint32_t* beta,
                        const uint32_t* gamma, const uint32_t* rho, uint32_t* z, uint32_t* e, uint32_t* s,
                        uint32_t* s1, uint32_t* s2) {
  env_t E; env_init(&E);
  mpz_t A, C, R, al, be, ga, ro, Z, U, W, Ee, S, S1, S2, t, gen;
  mpz_inits(A, C, R, al, be, ga, ro, Z, U, W, Ee, S, S1, S2, t, gen, NULL);
  for (int i = 0; i < batch; ++i) {
    env_load(&E, i, nkeys, N, nst, Nt, h1, h2, key_idx, st_idx);
    zin(A, a + (size_t)i * ORC_W256, ORC_W256);
    zin(C, cipher + (size_t)i * ORC_W4096, ORC_W4096);
    zin(R, r + (size_t)i * ORC_W2048, ORC_W2048);
    zin(al, alpha + (size_t)i * ORC_W768, ORC_W768);
    zin(be, beta + (size_t)i * ORC_W2048, ORC_W2048);
    zin(ga, gamma + (size_t)i * ORC_W2816, ORC_W2816);
    zin(ro, rho + (size_t)i * ORC_W2304, ORC_W2304);
    /* AliceZkpRound1::from :39-67 */
    mpz_powm(Z, E.h1, A, E.Nt); mpz_powm(t, E.h2, ro, E.Nt); mpz_mul(Z, Z, t); mpz_mod(Z, Z, E.Nt);          /* :52 */
    mpz_mul(U, al, E.N); mpz_add_ui(U, U, 1); mpz_powm(t, be, E.N, E.NN); mpz_mul(U, U, t); mpz_mod(U, U, E.NN); /* :53-55 */
    mpz_powm(W, E.h1, al, E.Nt); mpz_powm(t, E.h2, ga, E.Nt); mpz_mul(W, W, t); mpz_mod(W, W, E.Nt);        /* :56-57 */
    /* e = H(N, N+1, c, z, u, w) :175-182 */
    sha_t sh; sha_init(&sh);
    mpz_add_ui(gen, E.N, 1);
    chain_bigint(&sh, E.N); chain_bigint(&sh, gen); chain_bigint(&sh, C); chain_bigint(&sh, Z); chain_bigint(&sh, U); chain_bigint(&sh, W);
    result_bigint(&sh, Ee);
    /* AliceZkpRound2::from :78-90 */
    mpz_powm(S, R, Ee, E.N); mpz_mul(S, S, be); mpz_mod(S, S, E.N);
    mpz_mul(S1, Ee, A); mpz_add(S1, S1, al);
    mpz_mul(S2, Ee, ro); mpz_add(S2, S2, ga);
    zout(z + (size_t)i * ORC_W2048, ORC_W2048, Z);
    zout(e + (size_t)i * ORC_W256, ORC_W256, Ee);
    zout(s + (size_t)i * ORC_W2048, ORC_W2048, S);
    zout(s1 + (size_t)i * ORC_WS1, ORC_WS1, S1);
    zout(s2 + (size_t)i * ORC_WS2, ORC_WS2, S2);
  }
  mpz_clears(A, C, R, al, be, ga, ro, Z, U, W, Ee, S, S1, S2, t, gen, NULL);
  env_clear(&E);
}

void orc_alice_verify(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                      const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx, const uint32_t* cipher,
                      const uint32_t* z, const uint32_t* e, const uint32_t* s, const uint32_t* s1, const uint32_t* s2,
                      uint8_t* ok) {
  env_t E; env_init(&E);
  mpz_t C, Z, Ee, S, S1, S2, t, w, u, gs1, gen, e2;
  mpz_inits(C, Z, Ee, S, S1, S2, t, w, u, gs1, gen, e2, NULL);
  for (int i = 0; i < batch; ++i) {
    ok[i] = 0;
    env_load(&E, i, nkeys, N, nst, Nt, h1, h2, key_idx, st_idx);
    zin(C, cipher + (size_t)i * ORC_W4096, ORC_W4096);
    zin(Z, z + (size_t)i * ORC_W2048, ORC_W2048);
    zin(Ee, e + (size_t)i * ORC_W256, ORC_W256);
    zin(S, s + (size_t)i * ORC_W2048, ORC_W2048);
    zin(S1, s1 + (size_t)i * ORC_WS1, ORC_WS1);
    zin(S2, s2 + (size_t)i * ORC_WS2, ORC_WS2);
    if (mpz_cmp(S1, E.q3) > 0) continue;                                           /* :118 */
    mpz_powm(t, Z, Ee, E.Nt);
    if (!mpz_invert(t, t, E.Nt)) continue;                                          /* :122-127 */
    mpz_powm(w, E.h1, S1, E.Nt); mpz_powm(u, E.h2, S2, E.Nt);
    mpz_mul(w, w, u); mpz_mul(w, w, t); mpz_mod(w, w, E.Nt);                        /* :129-132 */
    mpz_mul(gs1, S1, E.N); mpz_add_ui(gs1, gs1, 1); mpz_mod(gs1, gs1, E.NN);       /* :134 */
    mpz_powm(t, C, Ee, E.NN);
    if (!mpz_invert(t, t, E.NN)) continue;                                          /* :135-139 */
    mpz_powm(u, S, E.N, E.NN); mpz_mul(u, u, gs1); mpz_mul(u, u, t); mpz_mod(u, u, E.NN); /* :141 */
    sha_t sh; sha_init(&sh);
    mpz_add_ui(gen, E.N, 1);
    chain_bigint(&sh, E.N); chain_bigint(&sh, gen); chain_bigint(&sh, C); chain_bigint(&sh, Z); chain_bigint(&sh, u); chain_bigint(&sh, w);
    result_bigint(&sh, e2);                                                          /* :143-150 */
    ok[i] = (uint8_t)(mpz_cmp(e2, Ee) == 0);
  }
  mpz_clears(C, Z, Ee, S, S1, S2, t, w, u, gs1, gen, e2, NULL);
  env_clear(&E);
}

/* ------------------------------------------------------------------------------------------ */
/* BobProof / BobProofExt (src/utilities/mta/range_proofs.rs:218-534)                          */
/* ------------------------------------------------------------------------------------------ */
static void bob_hash(mpz_t out, const env_t* E, const mpz_t a_enc, const mpz_t mta, const mpz_t z, const mpz_t zp,
                     const mpz_t t, const mpz_t v, const mpz_t w, const pt_t* X, const pt_t* u) {
  sha_t sh; sha_init(&sh);
  mpz_t gen; mpz_init(gen); mpz_add_ui(gen, E->N, 1);
  chain_bigint(&sh, E->N); chain_bigint(&sh, gen); chain_bigint(&sh, a_enc); chain_bigint(&sh, mta);
  chain_bigint(&sh, z); chain_bigint(&sh, zp); chain_bigint(&sh, t); chain_bigint(&sh, v); chain_bigint(&sh, w);
  if (X && u) {                                /* x_coord / y_coord as BigInt, :375-405 / :446-470 */
    chain_bigint(&sh, X->x); chain_bigint(&sh, X->y); chain_bigint(&sh, u->x); chain_bigint(&sh, u->y);
  }
  result_bigint(&sh, out);
  mpz_clear(gen);
}

void orc_bob_generate(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                      const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx, const uint32_t* a_enc,
                      const uint32_t* mta_enc, const uint32_t* b, const uint32_t* beta_prim, const uint32_t* r,
                      const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma, const uint32_t* rho,
                      const uint32_t* rho_prim, const uint32_t* sigma, const uint32_t* tau, int check, uint32_t* t,
                      uint32_t* z, uint32_t* e, uint32_t* s, uint32_t* s1, uint32_t* s2, uint32_t* t1, uint32_t* t2,
                      uint32_t* u) {
  env_t E; env_init(&E);
  mpz_t AE, ME, B, BP, R, al, be, ga, ro, rp, si, ta, Z, ZP, T, W, V, Ee, x, y;
  mpz_inits(AE, ME, B, BP, R, al, be, ga, ro, rp, si, ta, Z, ZP, T, W, V, Ee, x, y, NULL);
  pt_t G, X, U; pt_init(&G); pt_init(&X); pt_init(&U); pt_gen(&G);
  for (int i = 0; i < batch; ++i) {
    env_load(&E, i, nkeys, N, nst, Nt, h1, h2, key_idx, st_idx);
    zin(AE, a_enc + (size_t)i * ORC_W4096, ORC_W4096);
    zin(ME, mta_enc + (size_t)i * ORC_W4096, ORC_W4096);
    zin(B, b + (size_t)i * ORC_W256, ORC_W256);
    zin(BP, beta_prim + (size_t)i * ORC_W2048, ORC_W2048);
    zin(R, r + (size_t)i * ORC_W2048, ORC_W2048);
    zin(al, alpha + (size_t)i * ORC_W768, ORC_W768);
    zin(be, beta + (size_t)i * ORC_W2048, ORC_W2048);
    zin(ga, gamma + (size_t)i * ORC_W2560, ORC_W2560);
    zin(ro, rho + (size_t)i * ORC_W2304, ORC_W2304);
    zin(rp, rho_prim + (size_t)i * ORC_W2816, ORC_W2816);
    zin(si, sigma + (size_t)i * ORC_W2304, ORC_W2304);
    zin(ta, tau + (size_t)i * ORC_W2816, ORC_W2816);
    /* BobZkpRound1::from :238-249 */
    mpz_powm(Z, E.h1, B, E.Nt); mpz_powm(x, E.h2, ro, E.Nt); mpz_mul(Z, Z, x); mpz_mod(Z, Z, E.Nt);
    mpz_powm(ZP, E.h1, al, E.Nt); mpz_powm(x, E.h2, rp, E.Nt); mpz_mul(ZP, ZP, x); mpz_mod(ZP, ZP, E.Nt);
    mpz_powm(T, E.h1, BP, E.Nt); mpz_powm(x, E.h2, si, E.Nt); mpz_mul(T, T, x); mpz_mod(T, T, E.Nt);
    mpz_powm(W, E.h1, ga, E.Nt); mpz_powm(x, E.h2, ta, E.Nt); mpz_mul(W, W, x); mpz_mod(W, W, E.Nt);
    mpz_powm(V, AE, al, E.NN); mpz_mul(x, ga, E.N); mpz_add_ui(x, x, 1); mpz_mul(V, V, x);
    mpz_powm(y, be, E.N, E.NN); mpz_mul(V, V, y); mpz_mod(V, V, E.NN);
    if (check) {                                                                   /* :446-452 */
      pt_mul(&X, B, &G); pt_mul(&U, al, &G);
      bob_hash(Ee, &E, AE, ME, Z, ZP, T, V, W, &X, &U);
      if (u) pt_out(u + (size_t)i * ORC_WPOINT, &U);
    } else {
      bob_hash(Ee, &E, AE, ME, Z, ZP, T, V, W, NULL, NULL);
    }
    /* BobZkpRound2::from :281-297 */
    mpz_powm(x, R, Ee, E.N); mpz_mul(x, x, be); mpz_mod(x, x, E.N);
    zout(s + (size_t)i * ORC_W2048, ORC_W2048, x);
    mpz_mul(x, Ee, B); mpz_add(x, x, al);    zout(s1 + (size_t)i * ORC_WS1, ORC_WS1, x);
    mpz_mul(x, Ee, ro); mpz_add(x, x, rp);   zout(s2 + (size_t)i * ORC_WS2, ORC_WS2, x);
    mpz_mul(x, Ee, BP); mpz_add(x, x, ga);   zout(t1 + (size_t)i * ORC_WT1, ORC_WT1, x);
    mpz_mul(x, Ee, si); mpz_add(x, x, ta);   zout(t2 + (size_t)i * ORC_WS2, ORC_WS2, x);
    zout(t + (size_t)i * ORC_W2048, ORC_W2048, T);
    zout(z + (size_t)i * ORC_W2048, ORC_W2048, Z);
    zout(e + (size_t)i * ORC_W256, ORC_W256, Ee);
  }
  pt_clear(&G); pt_clear(&X); pt_clear(&U);
  mpz_clears(AE, ME, B, BP, R, al, be, ga, ro, rp, si, ta, Z, ZP, T, W, V, Ee, x, y, NULL);
  env_clear(&E);
}

void orc_bob_verify(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                    const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx, const uint32_t* a_enc,
                    const uint32_t* mta_enc, const uint32_t* t, const uint32_t* z, const uint32_t* e, const uint32_t* s,
                    const uint32_t* s1, const uint32_t* s2, const uint32_t* t1, const uint32_t* t2, const uint32_t* X,
                    const uint32_t* u, uint8_t* ok) {
  env_t E; env_init(&E);
  mpz_t AE, ME, T, Z, Ee, S, S1, S2, T1, T2, zi, mi, ti, zp, v, w, x, e2;
  mpz_inits(AE, ME, T, Z, Ee, S, S1, S2, T1, T2, zi, mi, ti, zp, v, w, x, e2, NULL);
  pt_t G, PX, PU, l, r2; pt_init(&G); pt_init(&PX); pt_init(&PU); pt_init(&l); pt_init(&r2); pt_gen(&G);
  for (int i = 0; i < batch; ++i) {
    ok[i] = 0;
    env_load(&E, i, nkeys, N, nst, Nt, h1, h2, key_idx, st_idx);
    zin(AE, a_enc + (size_t)i * ORC_W4096, ORC_W4096);
    zin(ME, mta_enc + (size_t)i * ORC_W4096, ORC_W4096);
    zin(T, t + (size_t)i * ORC_W2048, ORC_W2048);
    zin(Z, z + (size_t)i * ORC_W2048, ORC_W2048);
    zin(Ee, e + (size_t)i * ORC_W256, ORC_W256);
    zin(S, s + (size_t)i * ORC_W2048, ORC_W2048);
    zin(S1, s1 + (size_t)i * ORC_WS1, ORC_WS1);
    zin(S2, s2 + (size_t)i * ORC_WS2, ORC_WS2);
    zin(T1, t1 + (size_t)i * ORC_WT1, ORC_WT1);
    zin(T2, t2 + (size_t)i * ORC_WS2, ORC_WS2);
    if (mpz_cmp(S1, E.q3) > 0) continue;                                            /* :335 */
    mpz_powm(zi, Z, Ee, E.Nt); if (!mpz_invert(zi, zi, E.Nt)) continue;             /* :339-344 */
    mpz_powm(zp, E.h1, S1, E.Nt); mpz_powm(x, E.h2, S2, E.Nt); mpz_mul(zp, zp, x); mpz_mul(zp, zp, zi); mpz_mod(zp, zp, E.Nt);
    mpz_powm(mi, ME, Ee, E.NN); if (!mpz_invert(mi, mi, E.NN)) continue;            /* :351-355 */
    mpz_powm(v, AE, S1, E.NN); mpz_powm(x, S, E.N, E.NN); mpz_mul(v, v, x);
    mpz_mul(x, T1, E.N); mpz_add_ui(x, x, 1); mpz_mul(v, v, x); mpz_mul(v, v, mi); mpz_mod(v, v, E.NN); /* :357-361 */
    mpz_powm(ti, T, Ee, E.Nt); if (!mpz_invert(ti, ti, E.Nt)) continue;             /* :363-367 */
    mpz_powm(w, E.h1, T1, E.Nt); mpz_powm(x, E.h2, T2, E.Nt); mpz_mul(w, w, x); mpz_mul(w, w, ti); mpz_mod(w, w, E.Nt);
    if (X && u) {
      pt_in(&PX, X + (size_t)i * ORC_WPOINT); pt_in(&PU, u + (size_t)i * ORC_WPOINT);
      bob_hash(e2, &E, AE, ME, Z, zp, T, v, w, &PX, &PU);
    } else {
      bob_hash(e2, &E, AE, ME, Z, zp, T, v, w, NULL, NULL);
    }
    if (mpz_cmp(e2, Ee) != 0) continue;
    if (X && u) {                                                                   /* BobProofExt::verify :522-531 */
      pt_mul(&l, S1, &G);
      pt_mul(&r2, Ee, &PX); pt_add(&r2, &r2, &PU);
      if (!pt_eq(&l, &r2)) continue;
    }
    ok[i] = 1;
  }
  pt_clear(&G); pt_clear(&PX); pt_clear(&PU); pt_clear(&l); pt_clear(&r2);
  mpz_clears(AE, ME, T, Z, Ee, S, S1, S2, T1, T2, zi, mi, ti, zp, v, w, x, e2, NULL);
  env_clear(&E);
}

/* ------------------------------------------------------------------------------------------ */
/* PDLwSlackProof (src/utilities/zk_pdl_with_slack/mod.rs:68-179)                              */
/* ------------------------------------------------------------------------------------------ */
static void pdl_hash(mpz_t out, const pt_t* G, const pt_t* Q, const mpz_t c, const mpz_t z, const pt_t* u1,
                     const mpz_t u2, const mpz_t u3) {
  sha_t sh; sha_init(&sh);
  mpz_t t; mpz_init(t);
  pt_as_bigint(t, G); chain_bigint(&sh, t);
  pt_as_bigint(t, Q); chain_bigint(&sh, t);
  chain_bigint(&sh, c); chain_bigint(&sh, z);
  pt_as_bigint(t, u1); chain_bigint(&sh, t);
  chain_bigint(&sh, u2); chain_bigint(&sh, u3);
  result_bigint(&sh, out);
  mpz_clear(t);
}

void orc_pdl_prove(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                   const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx, const uint32_t* cipher,
                   const uint32_t* Q, const uint32_t* G, const uint32_t* x, const uint32_t* r, const uint32_t* alpha,
                   const uint32_t* beta, const uint32_t* rho, const uint32_t* gamma, uint32_t* z, uint32_t* u1,
                   uint32_t* u2, uint32_t* u3, uint32_t* s1, uint32_t* s2, uint32_t* s3) {
  env_t E; env_init(&E);
  mpz_t C, X, R, al, be, ro, ga, Z, U2, U3, Ee, t, one, np1;
  mpz_inits(C, X, R, al, be, ro, ga, Z, U2, U3, Ee, t, one, np1, NULL);
  mpz_set_ui(one, 1);
  pt_t PQ, PG, U1; pt_init(&PQ); pt_init(&PG); pt_init(&U1);
  for (int i = 0; i < batch; ++i) {
    env_load(&E, i, nkeys, N, nst, Nt, h1, h2, key_idx, st_idx);
    zin(C, cipher + (size_t)i * ORC_W4096, ORC_W4096);
    pt_in(&PQ, Q + (size_t)i * ORC_WPOINT); pt_in(&PG, G + (size_t)i * ORC_WPOINT);
    zin(X, x + (size_t)i * ORC_W256, ORC_W256);
    zin(R, r + (size_t)i * ORC_W2048, ORC_W2048);
    zin(al, alpha + (size_t)i * ORC_W768, ORC_W768);
    zin(be, beta + (size_t)i * ORC_W2048, ORC_W2048);
    zin(ro, rho + (size_t)i * ORC_W2304, ORC_W2304);
    zin(ga, gamma + (size_t)i * ORC_W2816, ORC_W2816);
    commit_unknown_order(Z, E.h1, E.h2, E.Nt, X, ro);                               /* :79-85 */
    pt_mul(&U1, al, &PG);                                                           /* :86 */
    mpz_add_ui(np1, E.N, 1);
    commit_unknown_order(U2, np1, be, E.NN, al, E.N);                               /* :87-93 */
    commit_unknown_order(U3, E.h1, E.h2, E.Nt, al, ga);                             /* :94-100 */
    pdl_hash(Ee, &PG, &PQ, C, Z, &U1, U2, U3);                                      /* :102-110 */
    mpz_mul(t, Ee, X); mpz_add(t, t, al); zout(s1 + (size_t)i * ORC_WS1, ORC_WS1, t);     /* :112 */
    commit_unknown_order(t, R, be, E.N, Ee, one); zout(s2 + (size_t)i * ORC_W2048, ORC_W2048, t); /* :113 */
    mpz_mul(t, Ee, ro); mpz_add(t, t, ga); zout(s3 + (size_t)i * ORC_WS2, ORC_WS2, t);    /* :114 */
    zout(z + (size_t)i * ORC_W2048, ORC_W2048, Z);
    pt_out(u1 + (size_t)i * ORC_WPOINT, &U1);
    zout(u2 + (size_t)i * ORC_W4096, ORC_W4096, U2);
    zout(u3 + (size_t)i * ORC_W2048, ORC_W2048, U3);
  }
  pt_clear(&PQ); pt_clear(&PG); pt_clear(&U1);
  mpz_clears(C, X, R, al, be, ro, ga, Z, U2, U3, Ee, t, one, np1, NULL);
  env_clear(&E);
}

void orc_pdl_verify(int batch, int nkeys, const uint32_t* N, int nst, const uint32_t* Nt, const uint32_t* h1,
                    const uint32_t* h2, const int32_t* key_idx, const int32_t* st_idx, const uint32_t* cipher,
                    const uint32_t* Q, const uint32_t* G, const uint32_t* z, const uint32_t* u1, const uint32_t* u2,
                    const uint32_t* u3, const uint32_t* s1, const uint32_t* s2, const uint32_t* s3, uint8_t* ok) {
  env_t E; env_init(&E);
  mpz_t C, Z, U2, U3, S1, S2, S3, Ee, t, t2, one, np1, ne;
  mpz_inits(C, Z, U2, U3, S1, S2, S3, Ee, t, t2, one, np1, ne, NULL);
  mpz_set_ui(one, 1);
  pt_t PQ, PG, U1, a, b; pt_init(&PQ); pt_init(&PG); pt_init(&U1); pt_init(&a); pt_init(&b);
  for (int i = 0; i < batch; ++i) {
    ok[i] = 0;
    env_load(&E, i, nkeys, N, nst, Nt, h1, h2, key_idx, st_idx);
    zin(C, cipher + (size_t)i * ORC_W4096, ORC_W4096);
    pt_in(&PQ, Q + (size_t)i * ORC_WPOINT); pt_in(&PG, G + (size_t)i * ORC_WPOINT); pt_in(&U1, u1 + (size_t)i * ORC_WPOINT);
    zin(Z, z + (size_t)i * ORC_W2048, ORC_W2048);
    zin(U2, u2 + (size_t)i * ORC_W4096, ORC_W4096);
    zin(U3, u3 + (size_t)i * ORC_W2048, ORC_W2048);
    zin(S1, s1 + (size_t)i * ORC_WS1, ORC_WS1);
    zin(S2, s2 + (size_t)i * ORC_W2048, ORC_W2048);
    zin(S3, s3 + (size_t)i * ORC_WS2, ORC_WS2);
    pdl_hash(Ee, &PG, &PQ, C, Z, &U1, U2, U3);                                      /* :128-136 */
    pt_mul(&a, S1, &PG);                                                            /* :138 */
    mpz_sub(t, E.q, Ee); pt_mul(&b, t, &PQ);                                        /* :139-141 */
    pt_add(&a, &a, &b);                                                             /* :142 */
    if (!pt_eq(&a, &U1)) continue;
    mpz_neg(ne, Ee);
    mpz_add_ui(np1, E.N, 1);
    /* the reference unwraps mod_inv here (panics on a non-invertible c or z); the oracle rejects */
    mpz_gcd(t, C, E.NN); if (mpz_cmp_ui(t, 1) != 0) continue;
    mpz_gcd(t, Z, E.Nt); if (mpz_cmp_ui(t, 1) != 0) continue;
    commit_unknown_order(t, np1, S2, E.NN, S1, E.N);                                /* :144-150 */
    commit_unknown_order(t2, t, C, E.NN, one, ne);                                  /* :151-157 */
    if (mpz_cmp(t2, U2) != 0) continue;
    commit_unknown_order(t, E.h1, E.h2, E.Nt, S1, S3);                              /* :159-165 */
    commit_unknown_order(t2, t, Z, E.Nt, one, ne);                                  /* :166-172 */
    if (mpz_cmp(t2, U3) != 0) continue;
    ok[i] = 1;
  }
  pt_clear(&PQ); pt_clear(&PG); pt_clear(&U1); pt_clear(&a); pt_clear(&b);
  mpz_clears(C, Z, U2, U3, S1, S2, S3, Ee, t, t2, one, np1, ne, NULL);
  env_clear(&E);
}

/* ------------------------------------------------------------------------------------------ */
/* curv DLogProof (App. A.3)                                                                   */
/* ------------------------------------------------------------------------------------------ */
static void dlog_challenge(mpz_t c, const pt_t* R, const pt_t* G, const pt_t* pk) {
  sha_t sh; sha_init(&sh);
  const pt_t* canon[3] = {R, G, pk};
  for (int i = 0; i < 3; ++i) chain_point(&sh, canon[ORC_ENC.ord_dlog[i] % 3]);
  result_bigint(&sh, c);
  mpz_mod(c, c, EC_Q);                                  /* result_scalar */
}
void orc_dlog_prove(int batch, const uint32_t* sk, const uint32_t* nonce, uint32_t* pk, uint32_t* R, uint32_t* z) {
  mpz_t s, k, c, t; mpz_inits(s, k, c, t, NULL);
  pt_t G, P, Rp; pt_init(&G); pt_init(&P); pt_init(&Rp); pt_gen(&G);
  for (int i = 0; i < batch; ++i) {
    zin(s, sk + (size_t)i * 8, 8); zin(k, nonce + (size_t)i * 8, 8);
    pt_mul(&Rp, k, &G); pt_mul(&P, s, &G);
    dlog_challenge(c, &Rp, &G, &P);
    mpz_mul(t, c, s); mpz_sub(t, k, t); mpz_mod(t, t, EC_Q);
    pt_out(pk + (size_t)i * 16, &P); pt_out(R + (size_t)i * 16, &Rp); zout(z + (size_t)i * 8, 8, t);
  }
  pt_clear(&G); pt_clear(&P); pt_clear(&Rp); mpz_clears(s, k, c, t, NULL);
}
void orc_dlog_verify(int batch, const uint32_t* pk, const uint32_t* R, const uint32_t* z, uint8_t* ok) {
  mpz_t c, zz; mpz_inits(c, zz, NULL);
  pt_t G, P, Rp, a, b; pt_init(&G); pt_init(&P); pt_init(&Rp); pt_init(&a); pt_init(&b); pt_gen(&G);
  for (int i = 0; i < batch; ++i) {
    pt_in(&P, pk + (size_t)i * 16); pt_in(&Rp, R + (size_t)i * 16); zin(zz, z + (size_t)i * 8, 8);
    dlog_challenge(c, &Rp, &G, &P);
    pt_mul(&a, zz, &G); pt_mul(&b, c, &P); pt_add(&a, &a, &b);
    ok[i] = (uint8_t)pt_eq(&a, &Rp);
  }
  pt_clear(&G); pt_clear(&P); pt_clear(&Rp); pt_clear(&a); pt_clear(&b); mpz_clears(c, zz, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* fixture helper: smallest prime > start (mpz_nextprime); used only to mint test key material */
/* (Paillier::keypair / generate_h1_h2_N_tilde draw random 1024-bit primes, party_i.rs:137-156) */
/* ------------------------------------------------------------------------------------------ */
void orc_nextprime(int k32, const uint32_t* start, uint32_t* out) {
  mpz_t x; mpz_init(x);
  zin(x, start, k32);
  mpz_nextprime(x, x);
  zout(out, k32, x);
  mpz_clear(x);
}

#include "gg20_oracle.c"
#include "lindell_oracle.c"
#include "sampler_oracle.c"
