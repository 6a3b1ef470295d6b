// 256-bit word helpers and the secp256k1 SCALAR field (mod q) on 8 x 32-bit words, one value per lane.
// Scalar arithmetic is cold next to the base field (mpe_fe.h): a handful of multiplications and one inversion per party
// and round.  Reduction folds 2^256 = QC (mod q) three times, straight-line.
// Compiles for the host too (MPE_FE_HOST) for tests/test_fe_cpu.py.
#pragma once
#include "mpe_fe.h"
#ifdef MPE_FE_HOST
#define MPE_CONST static const
#else
#define MPE_CONST __device__ __constant__ const
#endif

namespace mpe {
namespace ec {

MPE_CONST uint32_t FP[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                                 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
MPE_CONST uint32_t FQ[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                                                 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
// 2^256 - q (129 bits)
MPE_CONST uint32_t QC[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 0x1u};
MPE_CONST uint32_t GX[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu,
                                                 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
MPE_CONST uint32_t GY[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u,
                                                 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
// curv `Point::base_point2()`
MPE_CONST uint32_t H2X[8] = {0x0378b795u, 0xa8dc7bfau, 0x5ff3ce66u, 0xdd142e4bu,
                                                  0x4ba80116u, 0x34dd4521u, 0xe3a7326au, 0x08d13221u};
MPE_CONST uint32_t H2Y[8] = {0xf7c2be88u, 0x8217e9f7u, 0xdf0df07au, 0x807bcba1u,
                                                  0xbd565ea2u, 0x0848d50du, 0x77614b5cu, 0x5d41ac14u};

MPE_HD U256 u256_zero() { U256 r; for (int i = 0; i < 8; ++i) r.w[i] = 0; return r; }
MPE_HD U256 u256_one() { U256 r = u256_zero(); r.w[0] = 1; return r; }
MPE_HD U256 u256_load(const uint32_t* p) { U256 r; for (int i = 0; i < 8; ++i) r.w[i] = p[i]; return r; }
MPE_HD void u256_store(uint32_t* p, const U256& a) { for (int i = 0; i < 8; ++i) p[i] = a.w[i]; }
MPE_HD bool u256_is_zero(const U256& a) { uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= a.w[i]; return o == 0; }
MPE_HD bool u256_eq(const U256& a, const U256& b) { uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= a.w[i] ^ b.w[i]; return o == 0; }
MPE_HD bool u256_ge(const U256& a, const uint32_t* m) {
  for (int i = 7; i >= 0; --i) { if (a.w[i] != m[i]) return a.w[i] > m[i]; }
  return true;
}
MPE_HD uint32_t u256_add(U256& r, const U256& a, const U256& b) {
  uint64_t c = 0;
  for (int i = 0; i < 8; ++i) { c += (uint64_t)a.w[i] + b.w[i]; r.w[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)c;
}
MPE_HD uint32_t u256_sub_m(U256& r, const U256& a, const uint32_t* m) {
  int64_t c = 0;
  for (int i = 0; i < 8; ++i) { c += (int64_t)a.w[i] - (int64_t)m[i]; r.w[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)(c & 1);
}
MPE_HD uint32_t u256_sub(U256& r, const U256& a, const U256& b) { return u256_sub_m(r, a, b.w); }
MPE_HD void u256_add_m(U256& r, const U256& a, const uint32_t* m) {
  uint64_t c = 0;
  for (int i = 0; i < 8; ++i) { c += (uint64_t)a.w[i] + m[i]; r.w[i] = (uint32_t)c; c >>= 32; }
}
// 8x8 -> 16 words
MPE_HD void mul_wide(uint32_t (&t)[16], const U256& a, const U256& b) {
  for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t v = (uint64_t)a.w[i] * b.w[j] + t[i + j] + c;
      t[i + j] = (uint32_t)v;
      c = v >> 32;
    }
    t[i + 8] = (uint32_t)c;
  }
}

// ---- scalars: mod q ---------------------------------------------------------------------------
// 2^256 = QC (mod q), QC = 2^256 - q = 2^128 + QC[0..3] (129 bits).
// r = lo (8 words) + h (NH words) * QC, max(NH + 5, 9) words: straight-line, everything in registers.
template <int NH>
struct ScFold { static constexpr int NR = NH + 5 > 9 ? NH + 5 : 9; };
template <int NH>
MPE_HD void sc_fold(uint32_t (&r)[ScFold<NH>::NR], const uint32_t* lo, const uint32_t* h) {
  constexpr int NR = ScFold<NH>::NR;
  uint32_t p[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) p[i] = 0;
#pragma unroll
  for (int i = 0; i < NH; ++i) {                 // h * QC[0..3]
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t v = (uint64_t)h[i] * QC[j] + p[i + j] + c;
      p[i + j] = (uint32_t)v;
      c = v >> 32;
    }
    p[i + 4] = (uint32_t)c;
  }
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < NH; ++i) { c += (uint64_t)p[i + 4] + h[i]; p[i + 4] = (uint32_t)c; c >>= 32; }   // + h 2^128
  p[NH + 4] = (uint32_t)c;
  c = 0;
#pragma unroll
  for (int i = 0; i < NR; ++i) { c += (uint64_t)p[i] + (i < 8 ? lo[i] : 0u); r[i] = (uint32_t)c; c >>= 32; }
}
// 16 words -> [0, q)
MPE_HD U256 sc_reduce512(const uint32_t (&t)[16]) {
  uint32_t y[13], z[10], w[9];
  sc_fold<8>(y, t, t + 8);        // < 2^385 + 2^256
  sc_fold<5>(z, y, y + 8);        // y[8..12] < 2^130: < 2^260
  sc_fold<2>(w, z, z + 8);        // z[8..9] < 2^4:   < 2^256 + 2^134, w[8] in {0, 1}
  U256 r = u256_load(w);
  if (w[8]) u256_sub_m(r, r, FQ);                    // r + 2^256 - q: 2^256 + small - q < q, one subtraction settles it
  else if (u256_ge(r, FQ)) u256_sub_m(r, r, FQ);
  return r;
}
// 8 words: a value `Scalar::random()` can return — 0 < x < q
MPE_HD bool sc_is_canonical_nonzero(const uint32_t* x) {
  const U256 v = u256_load(x);
  uint32_t any = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) any |= v.w[i];
  return any != 0 && !u256_ge(v, FQ);
}
// an n-word integer mod q: Horner over 256-bit groups from the top
MPE_HD U256 sc_reduce(const uint32_t* x, int n) {
  const int top = n > 0 ? ((n - 1) >> 3) << 3 : 0;
  U256 acc;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc.w[i] = top + i < n ? x[top + i] : 0u;
  if (u256_ge(acc, FQ)) u256_sub_m(acc, acc, FQ);    // 2^256 < 2 q
#ifndef MPE_FE_HOST
#pragma unroll 1
#endif
  for (int g = top - 8; g >= 0; g -= 8) {
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { t[i] = x[g + i]; t[8 + i] = acc.w[i]; }
    acc = sc_reduce512(t);
  }
  return acc;
}
MPE_HD U256 sc_mul(const U256& a, const U256& b) { uint32_t t[16]; mul_wide(t, a, b); return sc_reduce512(t); }
MPE_HD U256 sc_add(const U256& a, const U256& b) {
  U256 r; const uint32_t c = u256_add(r, a, b);
  if (c || u256_ge(r, FQ)) u256_sub_m(r, r, FQ);
  return r;
}
MPE_HD U256 sc_sub(const U256& a, const U256& b) {
  U256 r; if (u256_sub(r, a, b)) u256_add_m(r, r, FQ);
  return r;
}
MPE_HD U256 sc_neg(const U256& a) { return u256_is_zero(a) ? a : sc_sub(u256_zero(), a); }
// a^(q-2) mod q: 4-bit windows of the (public) exponent
MPE_HDN U256 sc_inv(const U256& a) {
  U256 tab[16];
  tab[0] = u256_one();
  tab[1] = a;
#ifndef MPE_FE_HOST
#pragma unroll 1
#endif
  for (int i = 2; i < 16; ++i) tab[i] = sc_mul(tab[i - 1], a);
  uint32_t e[8];
  for (int i = 0; i < 8; ++i) e[i] = FQ[i];
  e[0] -= 2;
  U256 r = u256_one();
#ifndef MPE_FE_HOST
#pragma unroll 1
#endif
  for (int wi = 63; wi >= 0; --wi) {
    r = sc_mul(r, r); r = sc_mul(r, r); r = sc_mul(r, r); r = sc_mul(r, r);
    const uint32_t d = (e[wi >> 3] >> ((wi & 7) * 4)) & 15u;
    if (d) r = sc_mul(r, tab[d]);
  }
  return r;
}


// ---- GLV: k = r1 + r2 lambda (mod q) with |r1|, |r2| < 2^128 ------------------------------------------------
// lambda^3 = 1 (mod q), lambda (x, y) = (beta x, y) with beta^3 = 1 (mod p).  Lattice basis (a1, b1), (a2, b2) with
// a_i + b_i lambda = 0 (mod q); c1 = round(b2 k / q), c2 = round(-b1 k / q) through the precomputed
// g1 = round(2^384 b2 / q), g2 = round(2^384 (-b1) / q); r2 = c1 (-b1) + c2 (-b2), r1 = k - r2 lambda.
// (tests/test_fe_cpu.py re-derives the constants from a1, b1, a2 and checks the 128-bit bound.)
MPE_CONST uint32_t GLV_LAMBDA[8] = {0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu, 0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu};
MPE_CONST uint32_t GLV_G1[8] = {0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};
MPE_CONST uint32_t GLV_G2[8] = {0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};
MPE_CONST uint32_t GLV_MB1[8] = {0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u, 0u, 0u, 0u, 0u};
MPE_CONST uint32_t GLV_MB2[8] = {0x3DB1562Cu, 0xD765CDA8u, 0x0774346Du, 0x8A280AC5u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
// limbs (10 x 26) of beta
MPE_CONST uint32_t GLV_BETA[8] = {0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu};

// round(k g / 2^384): the top four words of the 512-bit product plus the rounding bit
MPE_HD U256 sc_mul_shift384(const U256& k, const uint32_t* g) {
  uint32_t t[16];
  mul_wide(t, k, u256_load(g));
  U256 r = u256_zero();
  uint64_t c = t[11] >> 31;
  for (int i = 0; i < 4; ++i) { c += t[12 + i]; r.w[i] = (uint32_t)c; c >>= 32; }
  r.w[4] = (uint32_t)c;
  return r;
}
struct GlvSplit { U256 r1, r2; bool neg1, neg2; };      // k = (neg1 ? -r1 : r1) + (neg2 ? -r2 : r2) lambda, r1, r2 < 2^128
MPE_HDN GlvSplit sc_split_lambda(const U256& k) {
  const U256 c1 = sc_mul(sc_mul_shift384(k, GLV_G1), u256_load(GLV_MB1));
  const U256 c2 = sc_mul(sc_mul_shift384(k, GLV_G2), u256_load(GLV_MB2));
  GlvSplit s;
  s.r2 = sc_add(c1, c2);
  s.r1 = sc_sub(k, sc_mul(s.r2, u256_load(GLV_LAMBDA)));
  // the representative of smaller absolute value: words 4..7 all set <=> negative (|r| < 2^128 << q / 2)
  s.neg1 = (s.r1.w[7] >> 31) != 0;
  s.neg2 = (s.r2.w[7] >> 31) != 0;
  if (s.neg1) s.r1 = sc_neg(s.r1);
  if (s.neg2) s.r2 = sc_neg(s.r2);
  return s;
}

}  // namespace ec
}  // namespace mpe
