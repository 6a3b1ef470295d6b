/* TEST INFRASTRUCTURE — an independent third-party checker, never part of the product path.
 *
 * Thin batch glue over OpenSSL's libcrypto (1.1.1l in this image, /opt/conda): secp256k1 point multiplication and addition
 * (EC_POINT_mul / EC_POINT_add), ECDSA verification (ECDSA_do_verify) and SHA-256, on the word layout of the C-ABI
 * (little-endian u32 words; points x[8] | y[8], all-zero = infinity).  None of the arithmetic below is ours: this is the
 * pin SURVEY.md 8c/8d asks for ("verifies under OpenSSL") and the role libsecp256k1 plays in the reference's own check
 * (src/protocols/multi_party_ecdsa/gg_2020/test.rs:711-748 `check_sig`: an independent library verifies (r, s) under y).
 * Only tests/, __graft_entry__.smoke() and bench.py's post-timing check load it. */
#include <openssl/bn.h>
#include <openssl/crypto.h>
#include <openssl/ec.h>
#include <openssl/ecdsa.h>
#include <openssl/obj_mac.h>
#include <openssl/sha.h>
#include <stdint.h>
#include <string.h>

const char* ossl_version(void) { return OpenSSL_version(OPENSSL_VERSION); }

static void words_to_be(const uint32_t* w, int nw, unsigned char* out) {
  for (int i = 0; i < nw; ++i) {
    const uint32_t v = w[nw - 1 - i];
    out[4 * i] = (unsigned char)(v >> 24); out[4 * i + 1] = (unsigned char)(v >> 16);
    out[4 * i + 2] = (unsigned char)(v >> 8); out[4 * i + 3] = (unsigned char)v;
  }
}
static void be_to_words(const unsigned char* in, int nw, uint32_t* w) {
  for (int i = 0; i < nw; ++i)
    w[nw - 1 - i] = ((uint32_t)in[4 * i] << 24) | ((uint32_t)in[4 * i + 1] << 16) | ((uint32_t)in[4 * i + 2] << 8) | in[4 * i + 3];
}
static BIGNUM* bn_of(const uint32_t* w, int nw) {
  unsigned char b[64];
  words_to_be(w, nw, b);
  return BN_bin2bn(b, 4 * nw, NULL);
}
static int is_zero(const uint32_t* w, int nw) {
  uint32_t a = 0;
  for (int i = 0; i < nw; ++i) a |= w[i];
  return a == 0;
}
/* point words -> EC_POINT (returns 0 when the coordinates are not on the curve) */
static int point_of(const EC_GROUP* g, const uint32_t* p, EC_POINT* out, BN_CTX* ctx) {
  if (is_zero(p, 16)) return EC_POINT_set_to_infinity(g, out);
  BIGNUM *x = bn_of(p, 8), *y = bn_of(p + 8, 8);
  const int ok = EC_POINT_set_affine_coordinates(g, out, x, y, ctx);
  BN_free(x); BN_free(y);
  return ok;
}
static void point_out(const EC_GROUP* g, const EC_POINT* p, uint32_t* out, BN_CTX* ctx) {
  memset(out, 0, 64);
  if (EC_POINT_is_at_infinity(g, p)) return;
  BIGNUM *x = BN_new(), *y = BN_new();
  unsigned char b[32];
  if (EC_POINT_get_affine_coordinates(g, p, x, y, ctx)) {
    BN_bn2binpad(x, b, 32); be_to_words(b, 8, out);
    BN_bn2binpad(y, b, 32); be_to_words(b, 8, out + 8);
  }
  BN_free(x); BN_free(y);
}

/* out[i] = k[i] * G   (P == NULL)   or   k[i] * P[i] */
int ossl_ec_mul(int batch, const uint32_t* k, const uint32_t* P, uint32_t* out) {
  EC_GROUP* g = EC_GROUP_new_by_curve_name(NID_secp256k1);
  BN_CTX* ctx = BN_CTX_new();
  EC_POINT *r = EC_POINT_new(g), *q = EC_POINT_new(g);
  int bad = 0;
  for (int i = 0; i < batch; ++i) {
    BIGNUM* s = bn_of(k + (size_t)i * 8, 8);
    int ok;
    if (P) ok = point_of(g, P + (size_t)i * 16, q, ctx) && EC_POINT_mul(g, r, NULL, q, s, ctx);
    else ok = EC_POINT_mul(g, r, s, NULL, NULL, ctx);
    if (ok) point_out(g, r, out + (size_t)i * 16, ctx); else { memset(out + (size_t)i * 16, 0xff, 64); ++bad; }
    BN_free(s);
  }
  EC_POINT_free(r); EC_POINT_free(q); BN_CTX_free(ctx); EC_GROUP_free(g);
  return bad;
}

int ossl_ec_add(int batch, const uint32_t* P, const uint32_t* Q, uint32_t* out) {
  EC_GROUP* g = EC_GROUP_new_by_curve_name(NID_secp256k1);
  BN_CTX* ctx = BN_CTX_new();
  EC_POINT *a = EC_POINT_new(g), *b = EC_POINT_new(g), *r = EC_POINT_new(g);
  int bad = 0;
  for (int i = 0; i < batch; ++i) {
    const int ok = point_of(g, P + (size_t)i * 16, a, ctx) && point_of(g, Q + (size_t)i * 16, b, ctx) && EC_POINT_add(g, r, a, b, ctx);
    if (ok) point_out(g, r, out + (size_t)i * 16, ctx); else { memset(out + (size_t)i * 16, 0xff, 64); ++bad; }
  }
  EC_POINT_free(a); EC_POINT_free(b); EC_POINT_free(r); BN_CTX_free(ctx); EC_GROUP_free(g);
  return bad;
}

/* ok[i] = ECDSA_do_verify(digest = the 32-byte big-endian form of msg[i], (r[i], s[i]), public key pub[i * pub_stride]).
 * pub_stride = 0: one key for the whole batch.  The message IS the digest: the reference signs a 256-bit BigInt
 * (party_i.rs:850-936 `phase7_local_sig(.., message)`, verified with `verify(sig, y, message)`).  Returns the number accepted. */
int ossl_ecdsa_verify(int batch, const uint32_t* pub, int pub_stride, const uint32_t* msg, const uint32_t* r, const uint32_t* s,
                      uint8_t* ok) {
  EC_GROUP* g = EC_GROUP_new_by_curve_name(NID_secp256k1);
  BN_CTX* ctx = BN_CTX_new();
  EC_POINT* q = EC_POINT_new(g);
  EC_KEY* key = EC_KEY_new();
  EC_KEY_set_group(key, g);
  int accepted = 0, have_key = 0;
  for (int i = 0; i < batch; ++i) {
    ok[i] = 0;
    if (pub_stride || !have_key) {
      const uint32_t* pw = pub + (size_t)i * pub_stride;
      have_key = !is_zero(pw, 16) && point_of(g, pw, q, ctx) && EC_KEY_set_public_key(key, q);
      if (!have_key) { if (!pub_stride) break; continue; }
    }
    if (is_zero(r + (size_t)i * 8, 8) || is_zero(s + (size_t)i * 8, 8)) continue;
    ECDSA_SIG* sig = ECDSA_SIG_new();
    ECDSA_SIG_set0(sig, bn_of(r + (size_t)i * 8, 8), bn_of(s + (size_t)i * 8, 8));
    unsigned char dg[32];
    words_to_be(msg + (size_t)i * 8, 8, dg);
    if (ECDSA_do_verify(dg, 32, sig, key) == 1) { ok[i] = 1; ++accepted; }
    ECDSA_SIG_free(sig);
  }
  EC_KEY_free(key); EC_POINT_free(q); BN_CTX_free(ctx); EC_GROUP_free(g);
  return accepted;
}

void ossl_sha256(const unsigned char* data, size_t len, unsigned char* out32) { SHA256(data, len, out32); }

/* out = base^exp mod m on big-endian byte strings (BN_mod_exp): a second bignum engine beside GMP for spot checks */
int ossl_modexp(const unsigned char* base, int blen, const unsigned char* exp, int elen, const unsigned char* mod, int mlen,
                unsigned char* out) {
  BN_CTX* ctx = BN_CTX_new();
  BIGNUM *b = BN_bin2bn(base, blen, NULL), *e = BN_bin2bn(exp, elen, NULL), *m = BN_bin2bn(mod, mlen, NULL), *r = BN_new();
  const int ok = BN_mod_exp(r, b, e, m, ctx) && BN_bn2binpad(r, out, mlen) == mlen;
  BN_free(b); BN_free(e); BN_free(m); BN_free(r); BN_CTX_free(ctx);
  return ok ? 0 : 1;
}
