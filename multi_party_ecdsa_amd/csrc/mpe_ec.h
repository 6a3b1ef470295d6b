// secp256k1 field / scalar / point arithmetic and SHA-256, one item per lane (plain u32[8] values).
//
// Replaces, for the GG20 hot path, what the reference reaches through curv-kzen's
// `Point<Secp256k1>` / `Scalar<Secp256k1>` (libsecp256k1 underneath) and `sha2::Sha256`:
// call sites src/utilities/mta/mod.rs:147-148,166-171, src/utilities/zk_pdl_with_slack/mod.rs:86,
// 102-110,138-142, src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:546-936.
// EC work is ~1 % of a signing session's multiplies, so these are plain
// per-lane routines: 8x8 schoolbook with 64-bit accumulators, Jacobian coordinates, a fixed 4-bit
// window ladder with a constant operation sequence (scalars on this path are secret).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpe {
namespace ec {

struct U256 { uint32_t w[8]; };

__device__ __constant__ const uint32_t FP[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                                 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
__device__ __constant__ const uint32_t FQ[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                                                 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
// 2^256 - q (129 bits)
__device__ __constant__ const uint32_t QC[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 0x1u};
__device__ __constant__ const uint32_t GX[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu,
                                                 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
__device__ __constant__ const uint32_t GY[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u,
                                                 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
// curv `Point::base_point2()`
__device__ __constant__ const uint32_t H2X[8] = {0x0378b795u, 0xa8dc7bfau, 0x5ff3ce66u, 0xdd142e4bu,
                                                  0x4ba80116u, 0x34dd4521u, 0xe3a7326au, 0x08d13221u};
__device__ __constant__ const uint32_t H2Y[8] = {0xf7c2be88u, 0x8217e9f7u, 0xdf0df07au, 0x807bcba1u,
                                                  0xbd565ea2u, 0x0848d50du, 0x77614b5cu, 0x5d41ac14u};

__device__ __forceinline__ U256 u256_zero() { U256 r; for (int i = 0; i < 8; ++i) r.w[i] = 0; return r; }
__device__ __forceinline__ U256 u256_one() { U256 r = u256_zero(); r.w[0] = 1; return r; }
__device__ __forceinline__ U256 u256_load(const uint32_t* p) { U256 r; for (int i = 0; i < 8; ++i) r.w[i] = p[i]; return r; }
__device__ __forceinline__ void u256_store(uint32_t* p, const U256& a) { for (int i = 0; i < 8; ++i) p[i] = a.w[i]; }
__device__ __forceinline__ bool u256_is_zero(const U256& a) { uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= a.w[i]; return o == 0; }
__device__ __forceinline__ bool u256_eq(const U256& a, const U256& b) { uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= a.w[i] ^ b.w[i]; return o == 0; }
__device__ __forceinline__ bool u256_ge(const U256& a, const uint32_t* m) {
  for (int i = 7; i >= 0; --i) { if (a.w[i] != m[i]) return a.w[i] > m[i]; }
  return true;
}
__device__ __forceinline__ uint32_t u256_add(U256& r, const U256& a, const U256& b) {
  uint64_t c = 0;
  for (int i = 0; i < 8; ++i) { c += (uint64_t)a.w[i] + b.w[i]; r.w[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)c;
}
__device__ __forceinline__ uint32_t u256_sub_m(U256& r, const U256& a, const uint32_t* m) {
  int64_t c = 0;
  for (int i = 0; i < 8; ++i) { c += (int64_t)a.w[i] - (int64_t)m[i]; r.w[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)(c & 1);
}
__device__ __forceinline__ uint32_t u256_sub(U256& r, const U256& a, const U256& b) { return u256_sub_m(r, a, b.w); }
__device__ __forceinline__ void u256_add_m(U256& r, const U256& a, const uint32_t* m) {
  uint64_t c = 0;
  for (int i = 0; i < 8; ++i) { c += (uint64_t)a.w[i] + m[i]; r.w[i] = (uint32_t)c; c >>= 32; }
}
// 8x8 -> 16 words
__device__ __forceinline__ void mul_wide(uint32_t (&t)[16], const U256& a, const U256& b) {
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

// a^2 -> 16 words: 28 cross products doubled + 8 squares (36 multiplies instead of 64)
__device__ __forceinline__ void sqr_wide(uint32_t (&t)[16], const U256& a) {
  for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = i + 1; j < 8; ++j) {
      const uint64_t v = (uint64_t)a.w[i] * a.w[j] + t[i + j] + c;
      t[i + j] = (uint32_t)v;
      c = v >> 32;
    }
    t[i + 8] = (uint32_t)c;
  }
  uint32_t top = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { const uint32_t nt = t[i] >> 31; t[i] = (t[i] << 1) | top; top = nt; }
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t sq = (uint64_t)a.w[i] * a.w[i];
    c += (uint64_t)t[2 * i] + (uint32_t)sq; t[2 * i] = (uint32_t)c; c >>= 32;
    c += (uint64_t)t[2 * i + 1] + (sq >> 32); t[2 * i + 1] = (uint32_t)c; c >>= 32;
  }
}

// ---- field: mod p = 2^256 - 2^32 - 977 ------------------------------------------------------
__device__ __forceinline__ U256 fe_reduce_wide(const uint32_t (&t)[16]) {
  // t = lo + hi 2^256,  2^256 = 2^32 + 977 (mod p):  r = lo + hi*977 + (hi << 32), twice
  uint32_t r[10];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint64_t lo = i < 8 ? t[i] : 0, h977 = i < 8 ? (uint64_t)t[8 + i] * 977u : 0, hs = i >= 1 ? t[8 + i - 1] : 0;
    c += lo + (h977 & 0xFFFFFFFFu) + hs;
    r[i] = (uint32_t)c;
    c = (c >> 32) + (h977 >> 32);
  }
  r[9] = (uint32_t)c;                           // value < 2^(256+34)
  // second fold: hi2 = r[8..9] (< 2^34)
  const uint64_t hi2 = (uint64_t)r[8] | ((uint64_t)r[9] << 32);
  U256 o;
  uint64_t k = hi2 * 977u;                      // < 2^44
  c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t add = 0;
    if (i == 0) add = k & 0xFFFFFFFFu;
    if (i == 1) add = (k >> 32) + (hi2 & 0xFFFFFFFFu);
    if (i == 2) add = hi2 >> 32;
    c += (uint64_t)r[i] + add;
    o.w[i] = (uint32_t)c;
    c >>= 32;
  }
  // c is 0 or 1: one more tiny fold (2^256 = 2^32 + 977)
  if (c) {
    uint64_t d = (uint64_t)o.w[0] + 977u;
    o.w[0] = (uint32_t)d; d >>= 32;
    d += (uint64_t)o.w[1] + 1u; o.w[1] = (uint32_t)d; d >>= 32;
    for (int i = 2; i < 8 && d; ++i) { d += o.w[i]; o.w[i] = (uint32_t)d; d >>= 32; }
  }
  if (u256_ge(o, FP)) u256_sub_m(o, o, FP);
  return o;
}
__device__ __forceinline__ U256 fe_mul(const U256& a, const U256& b) { uint32_t t[16]; mul_wide(t, a, b); return fe_reduce_wide(t); }
__device__ __forceinline__ U256 fe_sqr(const U256& a) { uint32_t t[16]; sqr_wide(t, a); return fe_reduce_wide(t); }
__device__ __forceinline__ U256 fe_add(const U256& a, const U256& b) {
  U256 r; const uint32_t c = u256_add(r, a, b);
  if (c || u256_ge(r, FP)) u256_sub_m(r, r, FP);
  return r;
}
__device__ __forceinline__ U256 fe_sub(const U256& a, const U256& b) {
  U256 r; if (u256_sub(r, a, b)) u256_add_m(r, r, FP);
  return r;
}
__device__ __forceinline__ U256 fe_neg(const U256& a) { return u256_is_zero(a) ? a : fe_sub(u256_zero(), a); }
__device__ inline U256 fe_pow(const U256& a, const uint32_t* e) {   // a^e, e: 8 words (public exponents only)
  U256 r = u256_one();
  for (int i = 255; i >= 0; --i) {
    r = fe_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = fe_mul(r, a);
  }
  return r;
}
__device__ inline U256 fe_sqrn(U256 x, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) x = fe_sqr(x);
  return x;
}
// a^(p-2): p - 2 = 1^223 0 1^22 0000 101101 in binary; addition chain with 255 squarings + 15 multiplications
__device__ inline U256 fe_inv(const U256& a) {
  const U256 x2 = fe_mul(fe_sqr(a), a), x3 = fe_mul(fe_sqr(x2), a);
  const U256 x6 = fe_mul(fe_sqrn(x3, 3), x3), x9 = fe_mul(fe_sqrn(x6, 3), x3), x11 = fe_mul(fe_sqrn(x9, 2), x2);
  const U256 x22 = fe_mul(fe_sqrn(x11, 11), x11), x44 = fe_mul(fe_sqrn(x22, 22), x22);
  const U256 x88 = fe_mul(fe_sqrn(x44, 44), x44), x176 = fe_mul(fe_sqrn(x88, 88), x88);
  const U256 x220 = fe_mul(fe_sqrn(x176, 44), x44), x223 = fe_mul(fe_sqrn(x220, 3), x3);
  U256 t = fe_mul(fe_sqrn(x223, 23), x22);
  t = fe_mul(fe_sqrn(t, 5), a);
  t = fe_mul(fe_sqrn(t, 3), x2);
  return fe_mul(fe_sqrn(t, 2), a);
}

// ---- scalars: mod q ---------------------------------------------------------------------------
// reduce an n-word integer mod q by folding 2^256 = QC (mod q)
__device__ inline U256 sc_reduce(const uint32_t* x, int n) {
  // work buffer: fold from the top down to 8 words (+ small overflow)
  uint32_t buf[90];
  for (int i = 0; i < n; ++i) buf[i] = x[i];
  for (int i = n; i < 90; ++i) buf[i] = 0;
  int len = n < 8 ? 8 : n;
  while (len > 8) {
    // take the top word w at position len-1 (>= 8): buf += w * QC << 32*(len-1-8), then drop it
    const uint32_t w = buf[len - 1];
    buf[len - 1] = 0;
    const int pos = len - 1 - 8;
    uint64_t c = 0;
    for (int j = 0; j < 5; ++j) {
      c += (uint64_t)w * QC[j] + buf[pos + j];
      buf[pos + j] = (uint32_t)c;
      c >>= 32;
    }
    for (int j = pos + 5; c && j < 90; ++j) { c += buf[j]; buf[j] = (uint32_t)c; c >>= 32; }
    // the carry may have re-populated word len-1 (only when pos+5 >= len-1, i.e. len <= 13): loop handles it
    while (len > 8 && buf[len - 1] == 0) --len;
  }
  U256 r = u256_load(buf);
  while (u256_ge(r, FQ)) u256_sub_m(r, r, FQ);
  return r;
}
__device__ inline U256 sc_mul(const U256& a, const U256& b) { uint32_t t[16]; mul_wide(t, a, b); return sc_reduce(t, 16); }
__device__ inline U256 sc_add(const U256& a, const U256& b) {
  U256 r; const uint32_t c = u256_add(r, a, b);
  if (c || u256_ge(r, FQ)) u256_sub_m(r, r, FQ);
  return r;
}
__device__ inline U256 sc_sub(const U256& a, const U256& b) {
  U256 r; if (u256_sub(r, a, b)) u256_add_m(r, r, FQ);
  return r;
}
__device__ inline U256 sc_neg(const U256& a) { return u256_is_zero(a) ? a : sc_sub(u256_zero(), a); }
__device__ inline U256 sc_inv(const U256& a) {   // a^(q-2) mod q
  uint32_t e[8];
  for (int i = 0; i < 8; ++i) e[i] = FQ[i];
  e[0] -= 2;
  U256 r = u256_one();
  for (int i = 255; i >= 0; --i) {
    r = sc_mul(r, r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = sc_mul(r, a);
  }
  return r;
}

// ---- points ----------------------------------------------------------------------------------------
struct Aff { U256 x, y; bool inf; };
struct Jac { U256 x, y, z; };                  // z == 0 <=> infinity

__device__ __forceinline__ Jac jac_inf() { Jac r; r.x = u256_one(); r.y = u256_one(); r.z = u256_zero(); return r; }
__device__ __forceinline__ bool jac_is_inf(const Jac& p) { return u256_is_zero(p.z); }
__device__ __forceinline__ Jac jac_from_aff(const Aff& a) {
  if (a.inf) return jac_inf();
  Jac r; r.x = a.x; r.y = a.y; r.z = u256_one(); return r;
}
__device__ inline Jac jac_dbl(const Jac& p) {
  if (jac_is_inf(p) || u256_is_zero(p.y)) return jac_inf();
  // a = 0: dbl-2009-l
  const U256 A = fe_sqr(p.x), B = fe_sqr(p.y), Cc = fe_sqr(B);
  U256 D = fe_sub(fe_sqr(fe_add(p.x, B)), fe_add(A, Cc));
  D = fe_add(D, D);
  const U256 E = fe_add(fe_add(A, A), A), Fv = fe_sqr(E);
  Jac r;
  r.x = fe_sub(Fv, fe_add(D, D));
  U256 c8 = fe_add(Cc, Cc); c8 = fe_add(c8, c8); c8 = fe_add(c8, c8);
  r.y = fe_sub(fe_mul(E, fe_sub(D, r.x)), c8);
  r.z = fe_mul(fe_add(p.y, p.y), p.z);
  return r;
}
__device__ inline Jac jac_add(const Jac& p, const Jac& q) {
  if (jac_is_inf(p)) return q;
  if (jac_is_inf(q)) return p;
  const U256 z1z1 = fe_sqr(p.z), z2z2 = fe_sqr(q.z);
  const U256 u1 = fe_mul(p.x, z2z2), u2 = fe_mul(q.x, z1z1);
  const U256 s1 = fe_mul(fe_mul(p.y, q.z), z2z2), s2 = fe_mul(fe_mul(q.y, p.z), z1z1);
  const U256 h = fe_sub(u2, u1), rr = fe_sub(s2, s1);
  if (u256_is_zero(h)) return u256_is_zero(rr) ? jac_dbl(p) : jac_inf();
  const U256 hh = fe_sqr(h), hhh = fe_mul(h, hh), v = fe_mul(u1, hh);
  Jac r;
  r.x = fe_sub(fe_sub(fe_sqr(rr), hhh), fe_add(v, v));
  r.y = fe_sub(fe_mul(rr, fe_sub(v, r.x)), fe_mul(s1, hhh));
  r.z = fe_mul(fe_mul(p.z, q.z), h);
  return r;
}
// p + q with q affine (z = 1): 8 multiplications + 3 squarings
__device__ inline Jac jac_add_aff(const Jac& p, const Aff& q) {
  if (q.inf) return p;
  if (jac_is_inf(p)) { Jac r; r.x = q.x; r.y = q.y; r.z = u256_one(); return r; }
  const U256 z1z1 = fe_sqr(p.z);
  const U256 u2 = fe_mul(q.x, z1z1), s2 = fe_mul(fe_mul(q.y, p.z), z1z1);
  const U256 h = fe_sub(u2, p.x), rr = fe_sub(s2, p.y);
  if (u256_is_zero(h)) return u256_is_zero(rr) ? jac_dbl(p) : jac_inf();
  const U256 hh = fe_sqr(h), hhh = fe_mul(h, hh), v = fe_mul(p.x, hh);
  Jac r;
  r.x = fe_sub(fe_sub(fe_sqr(rr), hhh), fe_add(v, v));
  r.y = fe_sub(fe_mul(rr, fe_sub(v, r.x)), fe_mul(p.y, hhh));
  r.z = fe_mul(p.z, h);
  return r;
}
// equality without leaving projective coordinates (no inversion)
__device__ inline bool jac_eq_aff(const Jac& p, const Aff& a) {
  if (a.inf || jac_is_inf(p)) return a.inf && jac_is_inf(p);
  const U256 zz = fe_sqr(p.z);
  return u256_eq(p.x, fe_mul(a.x, zz)) && u256_eq(p.y, fe_mul(a.y, fe_mul(zz, p.z)));
}
__device__ inline bool jac_eq(const Jac& p, const Jac& q) {
  if (jac_is_inf(p) || jac_is_inf(q)) return jac_is_inf(p) && jac_is_inf(q);
  const U256 z1z1 = fe_sqr(p.z), z2z2 = fe_sqr(q.z);
  return u256_eq(fe_mul(p.x, z2z2), fe_mul(q.x, z1z1)) &&
         u256_eq(fe_mul(p.y, fe_mul(z2z2, q.z)), fe_mul(q.y, fe_mul(z1z1, p.z)));
}
__device__ inline Aff jac_to_aff(const Jac& p) {
  Aff a;
  if (jac_is_inf(p)) { a.inf = true; a.x = u256_zero(); a.y = u256_zero(); return a; }
  const U256 zi = fe_inv(p.z), zi2 = fe_sqr(zi);
  a.x = fe_mul(p.x, zi2);
  a.y = fe_mul(p.y, fe_mul(zi2, zi));
  a.inf = false;
  return a;
}
__device__ inline Aff aff_neg(const Aff& a) { Aff r = a; if (!a.inf) r.y = fe_neg(a.y); return r; }
// k*P, k already reduced mod q.  Fixed 4-bit windows, constant sequence of doublings and additions.
__device__ inline Jac jac_mul(const U256& k, const Aff& P) {
  Jac tab[16];
  tab[0] = jac_inf();
  tab[1] = jac_from_aff(P);
  for (int i = 2; i < 16; ++i) tab[i] = (i & 1) ? jac_add(tab[i - 1], tab[1]) : jac_dbl(tab[i >> 1]);
  Jac acc = jac_inf();
#pragma unroll 1
  for (int wi = 63; wi >= 0; --wi) {
    acc = jac_dbl(jac_dbl(jac_dbl(jac_dbl(acc))));
    const uint32_t d = (k.w[wi >> 3] >> ((wi & 7) * 4)) & 15u;
    acc = jac_add(acc, tab[d]);
  }
  return acc;
}
__device__ __forceinline__ Aff aff_gen() { Aff g; g.x = u256_load(GX); g.y = u256_load(GY); g.inf = false; return g; }
__device__ __forceinline__ Aff aff_h2() { Aff g; g.x = u256_load(H2X); g.y = u256_load(H2Y); g.inf = false; return g; }
// Comb tables of the two fixed generators: COMB[g][w][d-1] = d * 16^w * (g == 0 ? G : base_point2), affine x|y,
// d = 1..15, w = 0..63 (123 KB, filled once per device by ec_comb_build_kernel when a context is created).
// k*G is then 64 mixed additions and no doublings; the additions always run (digit 0 adds a dummy and
// keeps the old accumulator), so the operation sequence does not depend on the scalar.
__device__ uint32_t COMB[2][64][15][16];
__global__ void __launch_bounds__(64) ec_comb_build_kernel() {
  const int w = threadIdx.x & 63, g = blockIdx.x;
  if (g > 1) return;
  Jac b = jac_from_aff(g ? aff_h2() : aff_gen());
  for (int i = 0; i < 4 * w; ++i) b = jac_dbl(b);
  const Aff ba = jac_to_aff(b);
  Jac acc = jac_from_aff(ba);
  for (int d = 1; d <= 15; ++d) {
    const Aff a = jac_to_aff(acc);
    for (int j = 0; j < 8; ++j) { COMB[g][w][d - 1][j] = a.x.w[j]; COMB[g][w][d - 1][8 + j] = a.y.w[j]; }
    acc = jac_add_aff(acc, ba);
  }
}
// k*G (g = 0) or k*base_point2 (g = 1), k already reduced mod q
__device__ __noinline__ Jac jac_mul_fixed(const U256& k, int g) {
  Jac acc = jac_inf();
#pragma unroll 1
  for (int w = 0; w < 64; ++w) {
    const uint32_t d = (k.w[w >> 3] >> ((w & 7) * 4)) & 15u;
    const uint32_t* e = COMB[g][w][d ? d - 1 : 0];
    Aff a; a.inf = false;
    for (int j = 0; j < 8; ++j) { a.x.w[j] = e[j]; a.y.w[j] = e[8 + j]; }
    const Jac sum = jac_add_aff(acc, a);
    if (d) acc = sum;
  }
  return acc;
}
__device__ __forceinline__ Jac jac_mul_gen(const U256& k) { return jac_mul_fixed(k, 0); }
__device__ __forceinline__ Jac jac_mul_h2(const U256& k) { return jac_mul_fixed(k, 1); }

// interface layout: x[8] | y[8], all-zero = infinity
__device__ __forceinline__ Aff aff_load(const uint32_t* p) {
  Aff a; a.x = u256_load(p); a.y = u256_load(p + 8); a.inf = u256_is_zero(a.x) && u256_is_zero(a.y); return a;
}
__device__ __forceinline__ void aff_store(uint32_t* p, const Aff& a) {
  if (a.inf) { for (int i = 0; i < 16; ++i) p[i] = 0; return; }
  u256_store(p, a.x); u256_store(p + 8, a.y);
}
// A point that arrives from a peer: both coordinates canonical (< p), on the curve y^2 = x^3 + 7, not the point at infinity.
// curv's Point deserialisation performs this check in the reference; without it a secret scalar multiplied into a peer's
// point is open to invalid-curve / small-subgroup inputs.  (secp256k1 has cofactor 1: on-curve = in the group.)
__device__ inline bool aff_valid(const Aff& a) {
  if (a.inf || u256_ge(a.x, FP) || u256_ge(a.y, FP)) return false;
  U256 rhs = fe_mul(fe_sqr(a.x), a.x), seven = u256_zero();
  seven.w[0] = 7;
  rhs = fe_add(rhs, seven);
  return u256_eq(fe_sqr(a.y), rhs);
}
__device__ __forceinline__ bool aff_eq(const Aff& a, const Aff& b) {
  if (a.inf || b.inf) return a.inf && b.inf;
  return u256_eq(a.x, b.x) && u256_eq(a.y, b.y);
}

// ---- SHA-256 (FIPS 180-4), streaming, one hash per lane ------------------------------------------
struct Sha256 {
  uint32_t h[8];
  uint32_t buf[16];      // current block, big-endian words
  uint64_t len;          // bytes absorbed
};
__device__ __constant__ const uint32_t SHA_K[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,
  0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,
  0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,
  0x06ca6351,0x14292967,0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
  0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,0x19a4c116,0x1e376c08,
  0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,
  0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__device__ inline void sha_init(Sha256& s) {
  s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
  s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
  for (int i = 0; i < 16; ++i) s.buf[i] = 0;
  s.len = 0;
}
__device__ __noinline__ void sha_block(Sha256& s) {     // out of line: every sha_byte site would otherwise carry a copy
  uint32_t w[64];
  for (int i = 0; i < 16; ++i) w[i] = s.buf[i];
  for (int i = 16; i < 64; ++i) {
    const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
  for (int i = 0; i < 64; ++i) {
    const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
    const uint32_t t1 = h + S1 + ch + SHA_K[i] + w[i];
    const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
    const uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d; s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
  for (int i = 0; i < 16; ++i) s.buf[i] = 0;
}
__device__ inline void sha_byte(Sha256& s, uint32_t byte) {
  const int off = (int)(s.len & 63);
  s.buf[off >> 2] |= (byte & 0xFFu) << (24 - 8 * (off & 3));
  s.len++;
  if ((s.len & 63) == 0) sha_block(s);
}
// DigestExt::chain_bigint: big-endian magnitude, minimal length (0 -> one 0x00 byte)
__device__ inline void sha_bigint(Sha256& s, const uint32_t* x, int nwords) {
  int top = nwords - 1;
  while (top > 0 && x[top] == 0) --top;
  int nb = 4;
  const uint32_t tw = x[top];
  if (tw < (1u << 8)) nb = 1; else if (tw < (1u << 16)) nb = 2; else if (tw < (1u << 24)) nb = 3;
  for (int b = nb - 1; b >= 0; --b) sha_byte(s, tw >> (8 * b));
  for (int i = top - 1; i >= 0; --i) {
    const uint32_t w = x[i];
    sha_byte(s, w >> 24); sha_byte(s, w >> 16); sha_byte(s, w >> 8); sha_byte(s, w);
  }
}
// bytes of a field element / coordinate, fixed 32 bytes big-endian
__device__ inline void sha_be32(Sha256& s, const U256& v) {
  for (int i = 7; i >= 0; --i) { const uint32_t w = v.w[i]; sha_byte(s, w >> 24); sha_byte(s, w >> 16); sha_byte(s, w >> 8); sha_byte(s, w); }
}
// Point::to_bytes(true) as hashed by zk_pdl_with_slack (BigInt::from_bytes(33 bytes) -> to_bytes: identical bytes)
__device__ inline void sha_point_compressed(Sha256& s, const Aff& p) { sha_byte(s, 2u + (p.y.w[0] & 1u)); sha_be32(s, p.x); }
// DigestExt::chain_point: Point::to_bytes(false), 65 bytes  [SURVEY.md App. A.2, recalled]
__device__ inline void sha_point_uncompressed(Sha256& s, const Aff& p) { sha_byte(s, 4u); sha_be32(s, p.x); sha_be32(s, p.y); }
// digest as 8 little-endian interface words (result_bigint)
__device__ inline U256 sha_final(Sha256& s) {
  const uint64_t bits = s.len * 8;
  sha_byte(s, 0x80);
  while ((s.len & 63) != 56) sha_byte(s, 0);
  for (int i = 7; i >= 0; --i) sha_byte(s, (uint32_t)(bits >> (8 * i)));
  U256 r;
  for (int i = 0; i < 8; ++i) r.w[i] = s.h[7 - i];
  return r;
}

}  // namespace ec
}  // namespace mpe
