// Host build of the device field / point code (mpe_fe.h, mpe_jac.h with MPE_FE_HOST) behind a C interface, so that
// tests/test_fe_cpu.py can fuzz the exact limb algorithms against Python big integers without a GPU.
//   g++ -O2 -shared -fPIC -DMPE_FE_HOST -I multi_party_ecdsa_amd/csrc tools/model/fe_host.cpp -o build/libfe_host.so
#include "mpe_jac.h"
#include <string.h>
using namespace mpe::ec;

static Fe raw(const uint32_t* l) { Fe r; memcpy(r.n, l, 40); return r; }
static void out_words(uint32_t* o, const Fe& a) { const U256 w = fe_to_u256(fe_normalize(a)); memcpy(o, w.w, 32); }

extern "C" {
// limbs in (any magnitude the caller likes), canonical words out
void feh_mul(const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t* out_limbs) {
  const Fe r = fe_mul(raw(a), raw(b)); if (out_limbs) memcpy(out_limbs, r.n, 40); out_words(out, r); }
void feh_sqr(const uint32_t* a, uint32_t* out, uint32_t* out_limbs) {
  const Fe r = fe_sqr(raw(a)); if (out_limbs) memcpy(out_limbs, r.n, 40); out_words(out, r); }
void feh_weak(const uint32_t* a, uint32_t* out_limbs) { const Fe r = fe_weak(raw(a)); memcpy(out_limbs, r.n, 40); }
void feh_normalize(const uint32_t* a, uint32_t* out) { out_words(out, raw(a)); }
int feh_is_zero(const uint32_t* a) { return fe_is_zero(raw(a)) ? 1 : 0; }
void feh_neg(const uint32_t* a, uint32_t m, uint32_t* out_limbs) { const Fe r = fe_neg(raw(a), m); memcpy(out_limbs, r.n, 40); }
void feh_from_words(const uint32_t* w, uint32_t* out_limbs) { U256 u; memcpy(u.w, w, 32); const Fe r = fe_from_u256(u); memcpy(out_limbs, r.n, 40); }
void feh_inv(const uint32_t* a, uint32_t* out) { out_words(out, fe_inv(raw(a))); }

static Aff aff_in(const uint32_t* p) {
  Aff a; memcpy(a.x.w, p, 32); memcpy(a.y.w, p + 8, 32);
  uint32_t o = 0; for (int i = 0; i < 16; ++i) o |= p[i];
  a.inf = o == 0; return a; }
static void aff_out(uint32_t* p, const Aff& a) {
  if (a.inf) { memset(p, 0, 64); return; }
  memcpy(p, a.x.w, 32); memcpy(p + 8, a.y.w, 32); }
void ech_mul(const uint32_t* k, const uint32_t* P, uint32_t* out) { U256 kk; memcpy(kk.w, k, 32); aff_out(out, jac_to_aff(jac_mul(kk, aff_in(P)))); }
void ech_mul_w4(const uint32_t* k, const uint32_t* P, uint32_t* out) { U256 kk; memcpy(kk.w, k, 32); aff_out(out, jac_to_aff(jac_mul_w4(kk, aff_in(P)))); }
void ech_add(const uint32_t* P, const uint32_t* Q, uint32_t* out) { aff_out(out, jac_to_aff(jac_add_aff(jac_from_aff(aff_in(P)), aff_in(Q)))); }
// (a P) + (b Q) through the general Jacobian addition; a, b small
void ech_add_jac(const uint32_t* ka, const uint32_t* P, const uint32_t* kb, const uint32_t* Q, uint32_t* out) {
  U256 a, b; memcpy(a.w, ka, 32); memcpy(b.w, kb, 32);
  aff_out(out, jac_to_aff(jac_add(jac_mul(a, aff_in(P)), jac_mul(b, aff_in(Q))))); }
int ech_eq(const uint32_t* ka, const uint32_t* P, const uint32_t* Q) {
  U256 a; memcpy(a.w, ka, 32);
  const Jac j = jac_mul(a, aff_in(P));
  return (jac_eq_aff(j, aff_in(Q)) ? 1 : 0) | (jac_eq(j, jac_from_aff(aff_in(Q))) ? 2 : 0); }
int ech_on_curve(const uint32_t* P) { return aff_on_curve(aff_in(P)) ? 1 : 0; }
// comb table of B: tab[w][d-1] = d 16^w B as 20 limbs; then k B from it
void ech_comb_build(const uint32_t* B, uint32_t* tab) {
  const Aff b0 = aff_in(B);
  Jac b = jac_from_aff(b0);
  for (int w = 0; w < 64; ++w) {
    const Aff ba = jac_to_aff(b);
    const AffL bl = affl_from_aff(ba);
    Jac acc = jac_from_affl(bl);
    for (int d = 1; d <= 15; ++d) {
      const AffL e = affl_from_aff(jac_to_aff(acc));
      memcpy(tab + ((size_t)w * 15 + d - 1) * 20, e.x.n, 40); memcpy(tab + ((size_t)w * 15 + d - 1) * 20 + 10, e.y.n, 40);
      acc = jac_add_affl(acc, bl);
    }
    for (int i = 0; i < 4; ++i) b = jac_dbl(b);
  }
}
void ech_mul_comb(const uint32_t* k, const uint32_t* tab, uint32_t* out) { U256 kk; memcpy(kk.w, k, 32); aff_out(out, jac_to_aff(jac_mul_comb(kk, tab))); }
}

// ---- scalar field (mpe_sc.h) ----
extern "C" {
void sch_split(const uint32_t* k, uint32_t* r1, uint32_t* r2, int* neg) { U256 kk; memcpy(kk.w, k, 32); const GlvSplit s = sc_split_lambda(kk); memcpy(r1, s.r1.w, 32); memcpy(r2, s.r2.w, 32); neg[0] = s.neg1; neg[1] = s.neg2; }
void sch_reduce(const uint32_t* x, int n, uint32_t* out) { const U256 r = sc_reduce(x, n); memcpy(out, r.w, 32); }
void sch_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) { U256 x, y; memcpy(x.w, a, 32); memcpy(y.w, b, 32); const U256 r = sc_mul(x, y); memcpy(out, r.w, 32); }
void sch_inv(const uint32_t* a, uint32_t* out) { U256 x; memcpy(x.w, a, 32); const U256 r = sc_inv(x); memcpy(out, r.w, 32); }
}
