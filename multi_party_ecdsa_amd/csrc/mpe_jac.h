// secp256k1 points in Jacobian coordinates over the lazily reduced field of mpe_fe.h.
//
// Invariant of every Jac this file returns: X of magnitude 1, Y <= 3, Z <= 2 (see mpe_fe.h for "magnitude").
// Affine points at the interface (`Aff`) are canonical 8 x 32-bit words; table entries (`AffL`) are limbs of magnitude 1.
// Exceptional cases (infinity, equal / opposite operands) are handled exactly: verifier inputs are adversarial and the
// outputs must equal the reference's for every input.
// Compiles for the host too (MPE_FE_HOST): tests/test_fe_cpu.py runs whole scalar multiplications against the oracle.
#pragma once
#include "mpe_fe.h"
#include "mpe_sc.h"

namespace mpe {
namespace ec {

struct Aff { U256 x, y; bool inf; };
struct AffL { Fe x, y; };                      // never infinity
struct Jac { Fe x, y, z; bool inf; };

MPE_HD Jac jac_inf() { Jac r; r.x = fe_small(1); r.y = fe_small(1); r.z = fe_zero(); r.inf = true; return r; }
MPE_HD bool jac_is_inf(const Jac& p) { return p.inf; }
MPE_HD AffL affl_from_aff(const Aff& a) { AffL r; r.x = fe_from_u256(a.x); r.y = fe_from_u256(a.y); return r; }
MPE_HD Jac jac_from_affl(const AffL& a) { Jac r; r.x = a.x; r.y = a.y; r.z = fe_small(1); r.inf = false; return r; }
MPE_HD Jac jac_from_aff(const Aff& a) {
  if (a.inf) return jac_inf();
  return jac_from_affl(affl_from_aff(a));
}

// 2 P (a = 0): 3 M + 4 S.   XX = X^2, YY = Y^2, S = 4 X YY, M = 3 XX:  X3 = M^2 - 2 S, Y3 = M (S - X3) - 8 YY^2, Z3 = 2 Y Z
MPE_HD Jac jac_dbl(const Jac& p) {
  if (p.inf) return p;                                             // (no point of order two on secp256k1: Y != 0)
  const Fe xx = fe_sqr(p.x), yy = fe_sqr(p.y), yyyy = fe_sqr(yy);  // magnitudes 1
  const Fe s1 = fe_mul(p.x, yy);                                   // 1;  S = 4 s1
  const Fe m = fe_mul_int(xx, 3);                                  // 3
  Jac r;
  r.inf = false;
  r.x = fe_weak(fe_add(fe_sqr(m), fe_neg(fe_mul_int(s1, 8), 8)));  // 1 + 9 -> 1
  const Fe d = fe_add(fe_mul_int(s1, 4), fe_neg(r.x, 1));          // S - X3: 4 + 2 = 6
  r.y = fe_weak(fe_add(fe_mul(m, d), fe_neg(fe_mul_int(yyyy, 8), 8)));   // 1 + 9 -> 1
  r.z = fe_mul_int(fe_mul(p.y, p.z), 2);                           // 2
  return r;
}
// the tail shared by the additions: h = U2 - U1, rr = S2 - S1 (magnitudes <= 6), u1 (1), s1 (1), zz = Z1 [Z2] (<= 2)
MPE_HD Jac jac_add_tail(const Fe& h, const Fe& rr, const Fe& u1, const Fe& s1, const Fe& zz) {
  const Fe hh = fe_sqr(h), hhh = fe_mul(h, hh), v = fe_mul(u1, hh);        // 1, 1, 1
  Jac r;
  r.inf = false;
  r.x = fe_weak(fe_add(fe_add(fe_sqr(rr), fe_neg(hhh, 1)), fe_neg(fe_mul_int(v, 2), 2)));   // 1 + 2 + 3 -> 1
  r.y = fe_add(fe_mul(rr, fe_add(v, fe_neg(r.x, 1))), fe_neg(fe_mul(s1, hhh), 1));          // 1 + 2 = 3
  r.z = fe_mul(zz, h);                                                                      // 1
  return r;
}
MPE_HD Jac jac_add(const Jac& p, const Jac& q) {
  if (p.inf) return q;
  if (q.inf) return p;
  const Fe z1z1 = fe_sqr(p.z), z2z2 = fe_sqr(q.z);
  const Fe u1 = fe_mul(p.x, z2z2), u2 = fe_mul(q.x, z1z1);
  const Fe s1 = fe_mul(fe_mul(p.y, q.z), z2z2), s2 = fe_mul(fe_mul(q.y, p.z), z1z1);      // 3 x 2 = 6 <= 32
  const Fe h = fe_sub(u2, u1, 1), rr = fe_sub(s2, s1, 1);                                 // 3
  if (fe_is_zero(h)) return fe_is_zero(rr) ? jac_dbl(p) : jac_inf();
  return jac_add_tail(h, rr, u1, s1, fe_mul(p.z, q.z));
}
// p + q with q affine (Z2 = 1): 8 M + 3 S
MPE_HD Jac jac_add_affl(const Jac& p, const AffL& q) {
  if (p.inf) return jac_from_affl(q);
  const Fe z1z1 = fe_sqr(p.z);
  const Fe u2 = fe_mul(q.x, z1z1), s2 = fe_mul(fe_mul(q.y, p.z), z1z1);
  const Fe h = fe_sub(u2, p.x, 1), rr = fe_sub(s2, p.y, 3);                               // 1 + 2 = 3, 1 + 4 = 5
  if (fe_is_zero(h)) return fe_is_zero(rr) ? jac_dbl(p) : jac_inf();
  // u1 = X1 (1), s1 = Y1 (3): s1 * hhh = 3 x 1
  return jac_add_tail(h, rr, p.x, p.y, p.z);
}
MPE_HD Jac jac_add_aff(const Jac& p, const Aff& q) {
  if (q.inf) return p;
  return jac_add_affl(p, affl_from_aff(q));
}
MPE_HD Jac jac_neg(const Jac& p) { Jac r = p; r.y = fe_weak(fe_neg(p.y, 3)); return r; }
// equality without leaving projective coordinates (no inversion)
MPE_HD bool jac_eq_aff(const Jac& p, const Aff& a) {
  if (a.inf || p.inf) return a.inf && p.inf;
  const AffL q = affl_from_aff(a);
  const Fe zz = fe_sqr(p.z);
  return fe_eq(fe_mul(q.x, zz), p.x, 1) && fe_eq(fe_mul(q.y, fe_mul(zz, p.z)), p.y, 3);
}
MPE_HD bool jac_eq(const Jac& p, const Jac& q) {
  if (p.inf || q.inf) return p.inf && q.inf;
  const Fe z1z1 = fe_sqr(p.z), z2z2 = fe_sqr(q.z);
  return fe_eq(fe_mul(p.x, z2z2), fe_mul(q.x, z1z1), 1) &&
         fe_eq(fe_mul(p.y, fe_mul(z2z2, q.z)), fe_mul(q.y, fe_mul(z1z1, p.z)), 1);
}
MPE_HDN Aff jac_to_aff(const Jac& p) {
  Aff a;
  if (p.inf) { a.inf = true; for (int i = 0; i < 8; ++i) a.x.w[i] = a.y.w[i] = 0; return a; }
  const Fe zi = fe_inv(p.z), zi2 = fe_sqr(zi);
  a.x = fe_to_u256(fe_normalize(fe_mul(p.x, zi2)));
  a.y = fe_to_u256(fe_normalize(fe_mul(p.y, fe_mul(zi2, zi))));
  a.inf = false;
  return a;
}
// y^2 == x^3 + 7 for canonical words x, y < p
MPE_HD bool aff_on_curve(const Aff& a) {
  const AffL q = affl_from_aff(a);
  const Fe rhs = fe_add(fe_mul(fe_sqr(q.x), q.x), fe_small(7));
  return fe_eq(fe_sqr(q.y), rhs, 2);
}

// k P, k already reduced mod q: plain ladder, fixed 4-bit windows (kept as the cross-check of jac_mul in the host tests)
MPE_HDN Jac jac_mul_w4(const U256& k, const Aff& P) {
  if (P.inf) return jac_inf();
  const AffL pa = affl_from_aff(P);
  Jac tab[16];
  tab[0] = jac_inf();
  tab[1] = jac_from_affl(pa);
  for (int i = 2; i < 16; ++i) tab[i] = (i & 1) ? jac_add_affl(tab[i - 1], pa) : jac_dbl(tab[i >> 1]);
  Jac acc = jac_inf();
#ifndef MPE_FE_HOST
#pragma unroll 1
#endif
  for (int wi = 63; wi >= 0; --wi) {
    acc = jac_dbl(jac_dbl(jac_dbl(jac_dbl(acc))));
    const uint32_t d = (k.w[wi >> 3] >> ((wi & 7) * 4)) & 15u;
    acc = jac_add(acc, tab[d]);
  }
  return acc;
}
// signed base-32 digits of v < 2^129: v = sum d_i 32^i, d_i in [-16, 16], i < 26
MPE_HD void glv_recode(int8_t (&dg)[26], const U256& v) {
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 26; ++i) {
    const int bit = 5 * i, w = bit >> 5, sft = bit & 31;
    uint32_t d = v.w[w] >> sft;
    if (sft > 27) d |= v.w[w + 1] << (32 - sft);
    d = (d & 31u) + carry;
    carry = d > 16u ? 1u : 0u;
    dg[i] = (int8_t)((int)d - (int)(carry << 5));
  }
}
// k P, k already reduced mod q.  GLV: k = r1 + r2 lambda with 128-bit halves, lambda P = (beta X : Y : Z); one table of
// 1P..16P serves both halves.  26 signed 5-bit windows: 130 doublings + 52 additions instead of 256 + 64.  The sequence
// of doublings / additions / table reads is the same for every scalar (digit 0 adds a dummy entry and keeps the old
// accumulator; signs are selects).
MPE_HDN Jac jac_mul(const U256& k, const Aff& P) {
  if (P.inf) return jac_inf();
  const AffL pa = affl_from_aff(P);
  Jac tab[16];                                                    // tab[j] = (j + 1) P
  tab[0] = jac_from_affl(pa);
#ifndef MPE_FE_HOST
#pragma unroll 1
#endif
  for (int m = 1; m <= 8; ++m) {
    tab[2 * m - 1] = jac_dbl(tab[m - 1]);
    if (m < 8) tab[2 * m] = jac_add_affl(tab[2 * m - 1], pa);
  }
  const GlvSplit sp = sc_split_lambda(k);
  int8_t d1[26], d2[26];
  glv_recode(d1, sp.r1);
  glv_recode(d2, sp.r2);
  const Fe beta = fe_from_u256(u256_load(GLV_BETA));
  Jac acc = jac_inf();
#ifndef MPE_FE_HOST
#pragma unroll 1
#endif
  for (int wi = 25; wi >= 0; --wi) {
    acc = jac_dbl(jac_dbl(jac_dbl(jac_dbl(jac_dbl(acc)))));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int d = half ? d2[wi] : d1[wi];
      const int ad = d < 0 ? -d : d;
      Jac q = tab[ad ? ad - 1 : 0];
      const bool neg = (d < 0) != (half ? sp.neg2 : sp.neg1);
      const Fe yn = fe_weak(fe_neg(q.y, 3));
      for (int i = 0; i < 10; ++i) q.y.n[i] = neg ? yn.n[i] : q.y.n[i];
      if (half) q.x = fe_mul(q.x, beta);
      const Jac sum = jac_add(acc, q);
      if (ad) acc = sum;
    }
  }
  return acc;
}
// k B from a comb table: tab[w][d - 1] = d 16^w B as 20 limbs (x | y), d = 1..15, w = 0..63: 64 mixed additions, no
// doublings; the additions always run (digit 0 adds a dummy entry and keeps the old accumulator)
MPE_HDN Jac jac_mul_comb(const U256& k, const uint32_t* tab) {
  Jac acc = jac_inf();
#ifndef MPE_FE_HOST
#pragma unroll 1
#endif
  for (int w = 0; w < 64; ++w) {
    const uint32_t d = (k.w[w >> 3] >> ((w & 7) * 4)) & 15u;
    const uint32_t* e = tab + ((size_t)w * 15 + (d ? d - 1 : 0)) * 20;
    AffL a;
    for (int j = 0; j < 10; ++j) { a.x.n[j] = e[j]; a.y.n[j] = e[10 + j]; }
    const Jac sum = jac_add_affl(acc, a);
    if (d) acc = sum;
  }
  return acc;
}

}  // namespace ec
}  // namespace mpe
