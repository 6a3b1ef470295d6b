// secp256k1 base field, one element per lane, in 10 limbs of 26 bits with LAZY reduction.
//
// Why this shape on gfx950: `v_mad_u64_u32` (32x32 + 64 -> 64) issues at the same rate as an add-with-carry
// (profiles/r01_valu_rate.json), so a multiplication on saturated 32-bit words pays ~2 carry instructions per product.
// With 26-bit limbs a column of 10 products (< 2^52 each, times the operands' magnitudes) fits a 64-bit accumulator:
// 100 multiply-accumulates and no carry instruction inside the product, the reduction modulo
// p = 2^256 - 2^32 - 977 folded in column by column (2^260 = 2^36 + 0x3D10 mod p), and additions / negations are 10
// independent lane-local instructions without carry chains or compare-and-subtract.  Roughly half the instructions
// of the 8 x 32-bit schoolbook + fold it replaces.
//
// Magnitude discipline: an element of magnitude m has limbs n[0..8] <= 2 m (2^26 - 1) and n[9] <= 2 m (2^22 - 1).
// fe_mul / fe_sqr accept magnitudes with m1 * m2 <= 32 and return magnitude 1; fe_add adds magnitudes; fe_neg(a, m)
// returns magnitude m + 1.  Every limb stays below 2^32 up to magnitude 31.  The value is only canonical after
// fe_normalize.  (The same discipline as libsecp256k1's 10x26 field, which the reference reaches through curv; the
// code below is written for this kernel.)
//
// Compiles for the host too (MPE_FE_HOST) so that tests/test_fe_cpu.py can fuzz it against big-integer arithmetic.
#pragma once
#include <stdint.h>
#include <stddef.h>
// MPE_HD: small, always inlined.  MPE_HDN: whole ladders / inversions -- one out-of-line copy per translation unit (inlining
// them into every kernel that multiplies a point made mpe_lib.hip a ten-minute compile); their call overhead is nothing
// next to the ~10^5 instructions they run.
#ifdef MPE_FE_HOST
#define MPE_HD inline
#define MPE_HDN inline
#else
#include <hip/hip_runtime.h>
#define MPE_HD __device__ __forceinline__
#define MPE_HDN __device__ __noinline__
// every kernel that calls into the out-of-line EC functions asks for two waves per SIMD: the request propagates to the callees
// (AMDGPU attributor), which otherwise take all 512 registers and leave their callers at one wave
#define MPE_EC_OCC __attribute__((amdgpu_waves_per_eu(2)))
#endif

namespace mpe {
namespace ec {

struct U256 { uint32_t w[8]; };
struct Fe { uint32_t n[10]; };

constexpr uint32_t FE_M = 0x3FFFFFFu;      // 26 bits
constexpr uint32_t FE_MT = 0x3FFFFFu;      // 22 bits (top limb)
constexpr uint32_t FE_R0 = 0x3D10u;        // 2^260 = 2^36 + 0x3D10 (mod p):  R0 at limb 0 ...
constexpr uint32_t FE_R1S = 10;            // ... and 2^10 at limb 1
// limbs of p
MPE_HD uint32_t fe_p_limb(int i) { return i == 0 ? 0x3FFFC2Fu : i == 1 ? 0x3FFFFBFu : i == 9 ? FE_MT : FE_M; }

MPE_HD Fe fe_zero() { Fe r; for (int i = 0; i < 10; ++i) r.n[i] = 0; return r; }
MPE_HD Fe fe_small(uint32_t v) { Fe r = fe_zero(); r.n[0] = v; return r; }     // v < 2^26

// canonical 8 x 32-bit words (value < 2^256) -> limbs (magnitude 1)
MPE_HD Fe fe_from_u256(const U256& a) {
  Fe r;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int bit = 26 * i, w = bit >> 5, s = bit & 31;
    uint32_t v = a.w[w] >> s;
    if (s > 6 && w + 1 < 8) v |= a.w[w + 1] << (32 - s);
    r.n[i] = v & FE_M;
  }
  return r;
}
// a NORMALISED element -> words
MPE_HD U256 fe_to_u256(const Fe& a) {
  U256 r;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int bit = 32 * w, i = bit / 26, s = bit - 26 * i;      // word w starts inside limb i at bit s
    uint32_t v = a.n[i] >> s;
    int have = 26 - s;
    for (int j = i + 1; j < 10 && have < 32; ++j) { v |= a.n[j] << have; have += 26; }
    r.w[w] = v;
  }
  return r;
}

MPE_HD Fe fe_add(const Fe& a, const Fe& b) { Fe r; for (int i = 0; i < 10; ++i) r.n[i] = a.n[i] + b.n[i]; return r; }
// -a for a of magnitude <= m: 2 (m + 1) p - a, limb-wise non-negative; magnitude m + 1
MPE_HD Fe fe_neg(const Fe& a, uint32_t m) {
  Fe r;
  for (int i = 0; i < 10; ++i) r.n[i] = 2u * (m + 1u) * fe_p_limb(i) - a.n[i];
  return r;
}
// a - b, b of magnitude <= mb; magnitude ma + mb + 1
MPE_HD Fe fe_sub(const Fe& a, const Fe& b, uint32_t mb) {
  Fe r;
  for (int i = 0; i < 10; ++i) r.n[i] = a.n[i] + (2u * (mb + 1u) * fe_p_limb(i) - b.n[i]);
  return r;
}
MPE_HD Fe fe_mul_int(const Fe& a, uint32_t k) { Fe r; for (int i = 0; i < 10; ++i) r.n[i] = a.n[i] * k; return r; }

// the tail shared by fe_mul and fe_sqr: t[0..9] are 26-bit limbs, c is what is left at limb 10 (2^260)
MPE_HD Fe fe_fold_tail(uint32_t (&t)[10], uint64_t c) {
  const uint32_t u10 = (uint32_t)c & FE_M, u11 = (uint32_t)(c >> 26);      // c < 2^39: u11 < 2^13
  const uint32_t e = t[9] >> 22;                                           // the part of limb 9 at and above 2^256
  t[9] &= FE_MT;
  uint64_t x = (uint64_t)t[0] + (uint64_t)u10 * FE_R0 + (uint64_t)e * 0x3D1u;
  t[0] = (uint32_t)x & FE_M; x >>= 26;
  x += (uint64_t)t[1] + ((uint64_t)u10 << FE_R1S) + (uint64_t)u11 * FE_R0 + ((uint64_t)e << 6);
  t[1] = (uint32_t)x & FE_M; x >>= 26;
  x += (uint64_t)t[2] + ((uint64_t)u11 << FE_R1S);
  t[2] = (uint32_t)x & FE_M; x >>= 26;
  t[3] += (uint32_t)x;                                                     // <= 2^26 + 1
  Fe r;
  for (int i = 0; i < 10; ++i) r.n[i] = t[i];
  return r;
}

// a * b mod p (lazily reduced, magnitude 1).  Column k of the low half and column k + 10 of the high half are summed
// side by side; the high column is cut to 26 bits (u) and enters the low half as u R0 at limb k and u 2^10 at limb k+1.
MPE_HD Fe fe_mul(const Fe& a, const Fe& b) {
  uint64_t c = 0, d = 0;
  uint32_t t[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
#pragma unroll
    for (int i = k + 1; i < 10; ++i) d += (uint64_t)a.n[i] * b.n[k + 10 - i];
    const uint32_t u = (uint32_t)d & FE_M;
    d >>= 26;
#pragma unroll
    for (int i = 0; i <= k; ++i) c += (uint64_t)a.n[i] * b.n[k - i];
    c += (uint64_t)u * FE_R0;
    t[k] = (uint32_t)c & FE_M;
    c >>= 26;
    c += (uint64_t)u << FE_R1S;
  }
  // d == 0 here: the product is below 2^520 (magnitudes m1 m2 <= 32)
  return fe_fold_tail(t, c);
}
MPE_HD Fe fe_sqr(const Fe& a) {
  uint64_t c = 0, d = 0;
  uint32_t t[10], a2[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) a2[i] = a.n[i] << 1;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    // high column k + 10: pairs (i, j), i < j, i + j = k + 10, plus the square when k is even
#pragma unroll
    for (int i = k + 1; i < 10; ++i) {
      const int j = k + 10 - i;
      if (i < j) d += (uint64_t)a2[i] * a.n[j];
      else if (i == j) d += (uint64_t)a.n[i] * a.n[i];
    }
    const uint32_t u = (uint32_t)d & FE_M;
    d >>= 26;
#pragma unroll
    for (int i = 0; i <= k; ++i) {
      const int j = k - i;
      if (i < j) c += (uint64_t)a2[i] * a.n[j];
      else if (i == j) c += (uint64_t)a.n[i] * a.n[i];
    }
    c += (uint64_t)u * FE_R0;
    t[k] = (uint32_t)c & FE_M;
    c >>= 26;
    c += (uint64_t)u << FE_R1S;
  }
  return fe_fold_tail(t, c);
}

// carry pass: any magnitude (limbs < 2^32) -> magnitude 1, value unchanged mod p
MPE_HD Fe fe_weak(const Fe& a) {
  Fe r = a;
  const uint32_t e = r.n[9] >> 22;
  r.n[9] &= FE_MT;
  r.n[0] += e * 0x3D1u;                  // e < 2^10: no overflow for limbs < 2^32 - 2^20
  r.n[1] += e << 6;
  uint32_t cy = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { const uint32_t v = r.n[i] + cy; r.n[i] = v & FE_M; cy = v >> 26; }
  r.n[9] += cy;                          // <= 2^22 + 2^6
  return r;
}
// canonical representative in [0, p)
MPE_HD Fe fe_normalize(const Fe& a) {
  Fe r = fe_weak(a);                     // value < 2^256 + 2^33 or so; limbs 0..8 < 2^26, limb 9 may carry one extra bit
  // second pass: the (rare) bit above 2^256 left in limb 9
  uint32_t e = r.n[9] >> 22;
  r.n[9] &= FE_MT;
  uint32_t cy = e * 0x3D1u;
  { uint32_t v = r.n[0] + cy; r.n[0] = v & FE_M; cy = v >> 26; }
  { uint32_t v = r.n[1] + cy + (e << 6); r.n[1] = v & FE_M; cy = v >> 26; }
#pragma unroll
  for (int i = 2; i < 9; ++i) { const uint32_t v = r.n[i] + cy; r.n[i] = v & FE_M; cy = v >> 26; }
  r.n[9] += cy;                          // now value < 2^256 (limb 9 < 2^22) -- at most one subtraction of p is left
  // value >= p  <=>  value + (2^32 + 977) >= 2^256
  uint32_t s[10];
  cy = 0x3D1u;
  { uint32_t v = r.n[0] + cy; s[0] = v & FE_M; cy = v >> 26; }
  { uint32_t v = r.n[1] + cy + 0x40u; s[1] = v & FE_M; cy = v >> 26; }
#pragma unroll
  for (int i = 2; i < 9; ++i) { const uint32_t v = r.n[i] + cy; s[i] = v & FE_M; cy = v >> 26; }
  s[9] = r.n[9] + cy;
  const bool ge = (s[9] >> 22) != 0;
  s[9] &= FE_MT;
  if (ge) for (int i = 0; i < 10; ++i) r.n[i] = s[i];
  return r;
}
// is the value 0 modulo p?  (any magnitude)
MPE_HD bool fe_is_zero(const Fe& a) {
  const Fe r = fe_weak(a);               // value in [0, 2^256 + small): zero mod p iff it is 0 or p (2 p > 2^256 + small)
  uint32_t z0 = 0, z1 = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < 10; ++i) { z0 |= r.n[i]; z1 &= r.n[i] ^ ~fe_p_limb(i); }   // z1 stays all-ones iff every limb == p's
  return z0 == 0 || z1 == 0xFFFFFFFFu;
}
MPE_HD bool fe_eq(const Fe& a, const Fe& b, uint32_t mb) { return fe_is_zero(fe_sub(a, b, mb)); }

MPE_HD Fe fe_sqrn(Fe x, int n) {
#ifndef MPE_FE_HOST
#pragma unroll 1
#endif
  for (int i = 0; i < n; ++i) x = fe_sqr(x);
  return x;
}
// a^(p-2): p - 2 = 1^223 0 1^22 0000 101101 in binary; addition chain with 255 squarings + 15 multiplications.
// a of magnitude <= 5; result magnitude 1.
MPE_HDN Fe fe_inv(const Fe& a) {
  const Fe x2 = fe_mul(fe_sqr(a), a), x3 = fe_mul(fe_sqr(x2), a);
  const Fe x6 = fe_mul(fe_sqrn(x3, 3), x3), x9 = fe_mul(fe_sqrn(x6, 3), x3), x11 = fe_mul(fe_sqrn(x9, 2), x2);
  const Fe x22 = fe_mul(fe_sqrn(x11, 11), x11), x44 = fe_mul(fe_sqrn(x22, 22), x22);
  const Fe x88 = fe_mul(fe_sqrn(x44, 44), x44), x176 = fe_mul(fe_sqrn(x88, 88), x88);
  const Fe x220 = fe_mul(fe_sqrn(x176, 44), x44), x223 = fe_mul(fe_sqrn(x220, 3), x3);
  Fe t = fe_mul(fe_sqrn(x223, 23), x22);
  t = fe_mul(fe_sqrn(t, 5), a);
  t = fe_mul(fe_sqrn(t, 3), x2);
  return fe_mul(fe_sqrn(t, 2), a);
}

}  // namespace ec
}  // namespace mpe
