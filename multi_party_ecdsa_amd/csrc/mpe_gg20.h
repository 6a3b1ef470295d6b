// GG20 signing, all parties of a batch of sessions in lock-step on one GPU — the batched form of
// Round0..Round7 (src/protocols/multi_party_ecdsa/gg_2020/state_machine/sign/rounds.rs:67-692) over
// SignKeys / LocalSignature (src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:526-936) and
// MessageA / MessageB (src/utilities/mta/mod.rs:52-179).  This is what `round_based::dev::Simulation`
// does for one session in the reference's own test (state_machine/sign.rs:667-763), for B sessions at once:
// every round is a handful of batched launches over (session, party[, peer[, statement]]) items.
// Included by mpe_lib.hip.
//
// Index conventions (identical to oracle/gg20_oracle.c):
//   pi = b*S + i                      party instance (session b, signer ordinal i)
//   ap = pi*n + st                    Alice range proof of pi for statement st
//   pp = pi*(S-1) + jj                ordered pair (pi -> peer ordinal jj), ind = jj < i ? jj : jj+1 (rounds.rs:149)
//   mb = pp*2 + v                     MessageB of pi for that peer, v = 0 (gamma_i) / 1 (w_i)
#pragma once
#include <cstdio>
#include <cstdlib>

#include "mpe_proofs.h"

struct mpe_gg20_keys {
  int t = 0, n = 0, S = 0;
  int signers[8] = {0};
  mpe_paillier* pk = nullptr;       // n private Paillier keys (party a = key a)
  mpe_statements* stm = nullptr;    // n statements (party a = statement a)
  void* blob = nullptr;
  uint32_t* x = nullptr;            // [n][8]  key shares
  uint32_t* X = nullptr;            // [n][16] pk_vec
  uint32_t* y = nullptr;            // [16]    group public key
  int32_t* d_signers = nullptr;     // [S]
};

namespace mpe {
namespace gg {

struct Dim { int B, S, n, V, PV; };   // V: range-proof verifications per MessageB pair (2 faithful / 1 dedup); PV: PDL verifiers (S / 1)
__device__ __forceinline__ int ind_of(int i, int jj) { return jj < i ? jj : jj + 1; }
__device__ __forceinline__ int jme_of(int i, int ind) { return i < ind ? i : i - 1; }

// ---- index tables for the batched launches -------------------------------------------------------
struct Idx {
  int32_t *key_pi;                                  // [B*S]
  int32_t *pi_ap, *key_ap, *st_ap;                  // [B*S*n]
  int32_t *ap_vi, *pia_vi, *key_vi, *st_vi;         // [B*S*(S-1)*V*n]  verifier-side range-proof checks
  int32_t *pia_mb, *key_mb;                         // [B*P*2]
  int32_t *mbin_rv, *key_rv;                        // [B*P*2] receiver-ordered incoming MessageB
  int32_t *pi_pp, *key_pp, *st_pp;                  // [B*P]
  int32_t *pp_pv, *pip_pv, *key_pv, *st_pv;         // [B*PV*P] PDL verifications
};
__global__ void idx_kernel(Dim d, const int32_t* __restrict__ signers, Idx x, int total) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const int S = d.S, n = d.n, P1 = S - 1;
  if (g < d.B * S) { x.key_pi[g] = signers[g % S]; }
  if (g < d.B * S * n) { const int pi = g / n; x.pi_ap[g] = pi; x.key_ap[g] = signers[pi % S]; x.st_ap[g] = g % n; }
  if (g < d.B * S * P1 * d.V * n) {
    const int st = g % n, r1 = g / n, r2 = r1 / d.V, jj = r2 % P1, pi = r2 / P1, i = pi % S, b = pi / S;
    const int ind = ind_of(i, jj), pia = b * S + ind;
    x.ap_vi[g] = pia * n + st; x.pia_vi[g] = pia; x.key_vi[g] = signers[ind]; x.st_vi[g] = st;
  }
  if (g < d.B * S * P1 * 2) {
    const int pp = g >> 1, v = g & 1, jj = pp % P1, pi = pp / P1, i = pi % S, b = pi / S, ind = ind_of(i, jj);
    x.pia_mb[g] = b * S + ind; x.key_mb[g] = signers[ind];
    // as receiver pi, peer ordinal jj: the message that peer `ind` built for me
    x.mbin_rv[g] = (((b * S + ind) * P1) + jme_of(i, ind)) * 2 + v; x.key_rv[g] = signers[i];
  }
  if (g < d.B * S * P1) {
    const int jj = g % P1, pi = g / P1, i = pi % S;
    x.pi_pp[g] = pi; x.key_pp[g] = signers[i]; x.st_pp[g] = signers[ind_of(i, jj)];
  }
  if (g < d.B * d.PV * S * P1) {
    const int per_sess = S * P1, q = g % per_sess, r1 = g / per_sess, b = r1 / d.PV;   // verifier ordinal = r1 % PV (unused)
    const int pp = b * per_sess + q, jj = pp % P1, pi = pp / P1, i = pi % S;
    x.pp_pv[g] = pp; x.pip_pv[g] = pi; x.key_pv[g] = signers[i]; x.st_pv[g] = signers[ind_of(i, jj)];
  }
}

__global__ void gather_rows_kernel(int n, int words, const uint32_t* __restrict__ src, const int32_t* __restrict__ idx,
                                   uint32_t* __restrict__ out) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * words) return;
  const size_t j = g / words;
  out[g] = src[(size_t)idx[j] * words + (g - j * words)];
}

// ---- helpers -------------------------------------------------------------------------------------
__device__ inline ec::U256 lagrange0(const int32_t* signers, int S, int i) {
  ec::U256 num = ec::u256_one(), den = ec::u256_one();
  for (int j = 0; j < S; ++j) {
    if (j == i) continue;
    ec::U256 xj = ec::u256_zero(); xj.w[0] = (uint32_t)(signers[j] + 1);
    ec::U256 xi = ec::u256_zero(); xi.w[0] = (uint32_t)(signers[i] + 1);
    num = ec::sc_mul(num, xj);
    den = ec::sc_mul(den, ec::sc_sub(xj, xi));
  }
  return ec::sc_mul(num, ec::sc_inv(den));
}
// HashCommitment(compressed point as BigInt, blind)  (party_i.rs:577-580)
__device__ inline ec::U256 commit_point(const ec::Aff& P, const uint32_t* blind) {
  ec::Sha256 s; ec::sha_init(s);
  ec::sha_point_compressed(s, P);
  ec::sha_bigint(s, blind, 8);
  return ec::sha_final(s);
}
// (the points are passed as an array reference and the function is force-inlined: handing a pointer to a
//  lane-private array to an out-of-line function hung the kernel on gfx950 / ROCm 7.2)
template <int N>
__device__ __forceinline__ ec::U256 hash_points(const ec::Aff (&pts)[N]) {
  ec::Sha256 s; ec::sha_init(s);
#pragma unroll
  for (int i = 0; i < N; ++i) ec::sha_point_uncompressed(s, pts[i]);
  const ec::U256 d = ec::sha_final(s);
  return ec::sc_reduce(d.w, 8);
}
__device__ __forceinline__ ec::Aff mul_aff(const ec::U256& k, const ec::Aff& P) { return ec::jac_to_aff(ec::jac_mul(k, P)); }
__device__ __forceinline__ ec::Aff add_aff(const ec::Aff& a, const ec::Aff& b) {
  return ec::jac_to_aff(ec::jac_add(ec::jac_from_aff(a), ec::jac_from_aff(b)));
}

// ---- Round 0: SignKeys::create + phase1_broadcast (party_i.rs:546-589) ----------------------------
__global__ void __launch_bounds__(64) r0_kernel(Dim d, const int32_t* __restrict__ signers, const uint32_t* __restrict__ xs,
                          const uint32_t* __restrict__ Xs, const uint32_t* __restrict__ k_in, const uint32_t* __restrict__ gamma_in,
                          const uint32_t* __restrict__ blind, uint32_t* __restrict__ kq, uint32_t* __restrict__ gq,
                          uint32_t* __restrict__ w, uint32_t* __restrict__ k64, uint32_t* __restrict__ g_gamma,
                          uint32_t* __restrict__ g_w, uint32_t* __restrict__ com) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.S) return;
  const int i = pi % d.S;
  const ec::U256 k = ec::sc_reduce(k_in + (size_t)pi * 8, 8), g = ec::sc_reduce(gamma_in + (size_t)pi * 8, 8);
  const ec::U256 lam = lagrange0(signers, d.S, i);
  const ec::U256 wi = ec::sc_mul(lam, ec::sc_reduce(xs + (size_t)signers[i] * 8, 8));
  ec::u256_store(kq + (size_t)pi * 8, k);
  ec::u256_store(gq + (size_t)pi * 8, g);
  ec::u256_store(w + (size_t)pi * 8, wi);
  for (int j = 0; j < 64; ++j) k64[(size_t)pi * 64 + j] = j < 8 ? k.w[j] : 0u;
  const ec::Aff gg = ec::jac_to_aff(ec::jac_mul_gen(g));
  ec::aff_store(g_gamma + (size_t)pi * 16, gg);
  // g_w_vec[i] as the PEERS compute it, from pk_vec (SignKeys::g_w_vec, party_i.rs:527-544): lambda_i X_i — not from
  // the secret share, so that a share inconsistent with the public key is caught by the check of rounds.rs:281
  ec::aff_store(g_w + (size_t)pi * 16, ec::jac_to_aff(ec::jac_mul(lam, ec::aff_load(Xs + (size_t)signers[i] * 16))));
  ec::u256_store(com + (size_t)pi * 8, commit_point(gg, blind + (size_t)pi * 8));
}

// ---- Round 1 glue: per MessageB the multiplier b, beta_tag mod q, beta = -beta_tag (mta/mod.rs:132,146) ----
__global__ void mb_prep_kernel(Dim d, const uint32_t* __restrict__ gq, const uint32_t* __restrict__ w,
                               const uint32_t* __restrict__ beta_tag, uint32_t* __restrict__ bsel,
                               uint32_t* __restrict__ btq, uint32_t* __restrict__ beta) {
  const int mb = blockIdx.x * blockDim.x + threadIdx.x;
  if (mb >= d.B * d.S * (d.S - 1) * 2) return;
  const int pi = (mb >> 1) / (d.S - 1);
  const uint32_t* src = (mb & 1) ? w + (size_t)pi * 8 : gq + (size_t)pi * 8;
  for (int j = 0; j < 8; ++j) bsel[(size_t)mb * 8 + j] = src[j];
  const ec::U256 t = ec::sc_reduce(beta_tag + (size_t)mb * 64, 64);
  ec::u256_store(btq + (size_t)mb * 8, t);
  ec::u256_store(beta + (size_t)mb * 8, ec::sc_neg(t));
}

// ---- Round 2a: MessageB::verify_proofs_get_alpha after the decryption (mta/mod.rs:166-178, rounds.rs:281) ----
struct MsgB { const uint32_t *pk, *R, *z, *tpk, *tR, *tz; };   // [mb] b_proof / beta_tag_proof
__global__ void __launch_bounds__(64) r2a_kernel(Dim d, const int32_t* __restrict__ mbin_rv, const uint32_t* __restrict__ alpha_full,
                           const uint32_t* __restrict__ kq, MsgB m, const uint32_t* __restrict__ g_w,
                           uint32_t* __restrict__ alpha, uint8_t* __restrict__ ok) {
  const int rv = blockIdx.x * blockDim.x + threadIdx.x;
  const int P1 = d.S - 1;
  if (rv >= d.B * d.S * P1 * 2) return;
  const int v = rv & 1, pp = rv >> 1, jj = pp % P1, pi = pp / P1, i = pi % d.S, b = pi / d.S, ind = ind_of(i, jj);
  const int in = mbin_rv[rv];
  const ec::U256 al = ec::sc_reduce(alpha_full + (size_t)rv * 64, 64);
  ec::u256_store(alpha + (size_t)rv * 8, al);
  const ec::Aff Bpk = ec::aff_load(m.pk + (size_t)in * 16), BTpk = ec::aff_load(m.tpk + (size_t)in * 16);
  const ec::Jac g_alpha = ec::jac_mul_gen(al);
  const ec::Jac ba_btag = ec::jac_add_aff(ec::jac_mul(ec::u256_load(kq + (size_t)pi * 8), Bpk), BTpk);
  bool good = ec::jac_eq(g_alpha, ba_btag);
  {  // DLogProof::verify x2
    const ec::Aff R1 = ec::aff_load(m.R + (size_t)in * 16), R2 = ec::aff_load(m.tR + (size_t)in * 16);
    const ec::U256 c1 = dlog_challenge(R1, Bpk), c2 = dlog_challenge(R2, BTpk);
    const ec::Jac l1 = ec::jac_add(ec::jac_mul_gen(ec::sc_reduce(m.z + (size_t)in * 8, 8)), ec::jac_mul(c1, Bpk));
    const ec::Jac l2 = ec::jac_add(ec::jac_mul_gen(ec::sc_reduce(m.tz + (size_t)in * 8, 8)), ec::jac_mul(c2, BTpk));
    good = good && ec::jac_eq_aff(l1, R1) && ec::jac_eq_aff(l2, R2);
  }
  if (v == 1) good = good && ec::aff_eq(Bpk, ec::aff_load(g_w + (size_t)(b * d.S + ind) * 16));   // rounds.rs:281
  ok[rv] = good ? 1 : 0;
}

// ---- Round 2b: delta_i, sigma_i, T_i + PedersenProof::prove (party_i.rs:591-634) -------------------
struct Ped { uint32_t *T, *a1, *a2, *z1, *z2; };       // [pi]
__global__ void __launch_bounds__(64) r2b_kernel(Dim d, const uint32_t* __restrict__ kq, const uint32_t* __restrict__ gq,
                           const uint32_t* __restrict__ w, const uint32_t* __restrict__ alpha,
                           const uint32_t* __restrict__ beta, const uint32_t* __restrict__ l_in,
                           const uint32_t* __restrict__ s1_in, const uint32_t* __restrict__ s2_in,
                           uint32_t* __restrict__ delta_i, uint32_t* __restrict__ sigma_i, uint32_t* __restrict__ lq, Ped p) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.S) return;
  const int P1 = d.S - 1;
  const ec::U256 k = ec::u256_load(kq + (size_t)pi * 8);
  ec::U256 de = ec::sc_mul(k, ec::u256_load(gq + (size_t)pi * 8)), si = ec::sc_mul(k, ec::u256_load(w + (size_t)pi * 8));
  for (int jj = 0; jj < P1; ++jj) {
    const size_t m0 = ((size_t)pi * P1 + jj) * 2;
    de = ec::sc_add(de, ec::sc_add(ec::u256_load(alpha + m0 * 8), ec::u256_load(beta + m0 * 8)));
    si = ec::sc_add(si, ec::sc_add(ec::u256_load(alpha + (m0 + 1) * 8), ec::u256_load(beta + (m0 + 1) * 8)));
  }
  ec::u256_store(delta_i + (size_t)pi * 8, de);
  ec::u256_store(sigma_i + (size_t)pi * 8, si);
  const ec::U256 l = ec::sc_reduce(l_in + (size_t)pi * 8, 8);
  ec::u256_store(lq + (size_t)pi * 8, l);
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2();
  const ec::Aff T = ec::jac_to_aff(ec::jac_add(ec::jac_mul_gen(si), ec::jac_mul_h2(l)));
  const ec::U256 s1 = ec::sc_reduce(s1_in + (size_t)pi * 8, 8), s2 = ec::sc_reduce(s2_in + (size_t)pi * 8, 8);
  const ec::Aff a1 = ec::jac_to_aff(ec::jac_mul_gen(s1)), a2 = ec::jac_to_aff(ec::jac_mul_h2(s2));
  const ec::Aff hp[5] = {G, H, T, a1, a2};
  const ec::U256 e = hash_points(hp);
  ec::aff_store(p.T + (size_t)pi * 16, T);
  ec::aff_store(p.a1 + (size_t)pi * 16, a1);
  ec::aff_store(p.a2 + (size_t)pi * 16, a2);
  ec::u256_store(p.z1 + (size_t)pi * 8, ec::sc_add(s1, ec::sc_mul(e, si)));
  ec::u256_store(p.z2 + (size_t)pi * 8, ec::sc_add(s2, ec::sc_mul(e, l)));
}

// ---- Round 3: every party verifies every PedersenProof, reconstructs delta^-1 (rounds.rs:347-402) ---
__global__ void __launch_bounds__(64) r3_kernel(Dim d, const uint32_t* __restrict__ delta_i, Ped p, uint32_t* __restrict__ dinv, uint8_t* __restrict__ ok) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.S) return;
  const int b = pi / d.S;
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2();
  ec::U256 sum = ec::u256_zero();
  bool good = true;
  for (int j = 0; j < d.S; ++j) {
    const size_t o = (size_t)b * d.S + j;
    sum = ec::sc_add(sum, ec::u256_load(delta_i + o * 8));
    const ec::Aff T = ec::aff_load(p.T + o * 16), a1 = ec::aff_load(p.a1 + o * 16), a2 = ec::aff_load(p.a2 + o * 16);
    const ec::Aff hp[5] = {G, H, T, a1, a2};
    const ec::U256 e = hash_points(hp);
    const ec::Jac lhs = ec::jac_add(ec::jac_mul_gen(ec::u256_load(p.z1 + o * 8)), ec::jac_mul_h2(ec::u256_load(p.z2 + o * 8)));
    const ec::Jac rhs = ec::jac_add_aff(ec::jac_add_aff(ec::jac_mul(e, T), a1), a2);
    good = good && ec::jac_eq(lhs, rhs);
  }
  good = good && !ec::u256_is_zero(sum);
  ec::u256_store(dinv + (size_t)pi * 8, ec::sc_inv(sum));
  ok[pi] = good ? 1 : 0;
}

// ---- Round 4: phase4 -> R, R_dash (party_i.rs:642-687, rounds.rs:452) --------------------------------
__global__ void __launch_bounds__(64) r4_kernel(Dim d, const uint32_t* __restrict__ dinv, const uint32_t* __restrict__ g_gamma,
                          const uint32_t* __restrict__ com, const uint32_t* __restrict__ blind, const uint32_t* __restrict__ Bpk,
                          const uint32_t* __restrict__ kq, uint32_t* __restrict__ R, uint32_t* __restrict__ Rbar,
                          uint8_t* __restrict__ ok) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.S) return;
  const int P1 = d.S - 1, i = pi % d.S, b = pi / d.S;
  bool good = true;
  for (int jj = 0; jj < P1; ++jj) {
    const int ind = ind_of(i, jj);
    const size_t o = (size_t)b * d.S + ind;
    const ec::Aff gg = ec::aff_load(g_gamma + o * 16);
    const size_t in = ((o * P1) + jme_of(i, ind)) * 2;          // the gamma MessageB that `ind` sent me
    good = good && ec::aff_eq(ec::aff_load(Bpk + in * 16), gg);
    good = good && ec::u256_eq(commit_point(gg, blind + o * 8), ec::u256_load(com + o * 8));
  }
  ec::Jac acc = ec::jac_inf();
  for (int j = 0; j < d.S; ++j) acc = ec::jac_add(acc, ec::jac_from_aff(ec::aff_load(g_gamma + ((size_t)b * d.S + j) * 16)));
  const ec::Aff Rp = mul_aff(ec::u256_load(dinv + (size_t)pi * 8), ec::jac_to_aff(acc));
  ec::aff_store(R + (size_t)pi * 16, Rp);
  ec::aff_store(Rbar + (size_t)pi * 16, mul_aff(ec::u256_load(kq + (size_t)pi * 8), Rp));
  ok[pi] = good ? 1 : 0;
}

// ---- Round 5: R_dash sum, S_i and HomoELGamalProof::prove (party_i.rs:768-799) ------------------------
struct Heg { uint32_t *S, *T, *A3, *z1, *z2; };      // [pi]
__global__ void __launch_bounds__(64) r5_kernel(Dim d, const uint8_t* __restrict__ pdl_ok, const uint32_t* __restrict__ R,
                          const uint32_t* __restrict__ Rbar, const uint32_t* __restrict__ sigma_i, const uint32_t* __restrict__ lq,
                          const uint32_t* __restrict__ pedT, const uint32_t* __restrict__ s1_in, const uint32_t* __restrict__ s2_in,
                          Heg h, uint8_t* __restrict__ ok) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.S) return;
  const int P1 = d.S - 1, i = pi % d.S, b = pi / d.S, per = d.S * P1;
  bool good = true;
  // my PDL verifications: all S(S-1) proofs of the session (rounds.rs:546-558)
  const int vo = d.PV == 1 ? 0 : i;
  for (int q = 0; q < per; ++q) good = good && pdl_ok[((size_t)b * d.PV + vo) * per + q];
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2();
  ec::Jac acc = ec::jac_inf();
  for (int j = 0; j < d.S; ++j) acc = ec::jac_add(acc, ec::jac_from_aff(ec::aff_load(Rbar + ((size_t)b * d.S + j) * 16)));
  good = good && ec::jac_eq_aff(acc, G);                                                 // phase5_check_R_dash_sum
  const ec::Aff Rp = ec::aff_load(R + (size_t)pi * 16), T = ec::aff_load(pedT + (size_t)pi * 16);
  const ec::U256 si = ec::u256_load(sigma_i + (size_t)pi * 8), l = ec::u256_load(lq + (size_t)pi * 8);
  const ec::Aff Sp = mul_aff(si, Rp);
  const ec::U256 s1 = ec::sc_reduce(s1_in + (size_t)pi * 8, 8), s2 = ec::sc_reduce(s2_in + (size_t)pi * 8, 8);
  const ec::Aff A3 = mul_aff(s2, Rp), TT = ec::jac_to_aff(ec::jac_add(ec::jac_mul_h2(s1), ec::jac_mul_gen(s2)));
  const ec::Aff hp[7] = {TT, A3, Rp, H, G, T, Sp};
  const ec::U256 e = hash_points(hp);
  ec::aff_store(h.S + (size_t)pi * 16, Sp);
  ec::aff_store(h.T + (size_t)pi * 16, TT);
  ec::aff_store(h.A3 + (size_t)pi * 16, A3);
  ec::u256_store(h.z1 + (size_t)pi * 8, ec::u256_is_zero(l) ? s1 : ec::sc_add(s1, ec::sc_mul(l, e)));
  ec::u256_store(h.z2 + (size_t)pi * 8, ec::sc_add(s2, ec::sc_mul(si, e)));
  ok[pi] = good ? 1 : 0;
}

// ---- Round 6: every party verifies every HomoELGamalProof; sum S_i == y (party_i.rs:801-848) --------------
__global__ void __launch_bounds__(64) r6_kernel(Dim d, const uint32_t* __restrict__ R, const uint32_t* __restrict__ pedT, Heg h,
                          const uint32_t* __restrict__ y, uint8_t* __restrict__ ok) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.S) return;
  const int b = pi / d.S;
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2(), Rp = ec::aff_load(R + (size_t)pi * 16);
  bool good = true;
  ec::Jac acc = ec::jac_inf();
  for (int j = 0; j < d.S; ++j) {
    const size_t o = (size_t)b * d.S + j;
    const ec::Aff TT = ec::aff_load(h.T + o * 16), A3 = ec::aff_load(h.A3 + o * 16), D = ec::aff_load(pedT + o * 16),
                  E = ec::aff_load(h.S + o * 16);
    const ec::Aff hp[7] = {TT, A3, Rp, H, G, D, E};
    const ec::U256 e = hash_points(hp), z1 = ec::u256_load(h.z1 + o * 8), z2 = ec::u256_load(h.z2 + o * 8);
    const ec::Jac l1 = ec::jac_add(ec::jac_mul_h2(z1), ec::jac_mul_gen(z2));
    const ec::Jac r1 = ec::jac_add_aff(ec::jac_mul(e, D), TT);
    const ec::Jac l2 = ec::jac_mul(z2, Rp);
    const ec::Jac r2 = ec::jac_add_aff(ec::jac_mul(e, E), A3);
    good = good && ec::jac_eq(l1, r1) && ec::jac_eq(l2, r2);
    acc = ec::jac_add_aff(acc, E);
  }
  good = good && ec::jac_eq_aff(acc, ec::aff_load(y));
  ok[pi] = good ? 1 : 0;
}

// ---- small batches: the same checks with a group of adjacent lanes per item ---------------------------
// One party per lane runs an item's scalar multiplications back to back; with few sessions that leaves most SIMDs
// idle for the whole latency of the chain.  The *_group kernels give every item G lanes: the independent
// multiplications run one per lane (all lanes of a phase execute the same routine on different data), the Jacobian
// results meet in LDS and lane 0 of the group does the additions, comparisons and stores.  With many sessions the
// idle lanes of the groups cost more than the latency they hide (measured at 65 536 sessions: r2a 42 -> 57 ms,
// r3 22 -> 52 ms), so sign_chunk picks them only when the grouped launch still fits the chip.
struct JacSlots { ec::Jac v[64]; };

__global__ void __launch_bounds__(64) r2a_group_kernel(Dim d, const int32_t* __restrict__ mbin_rv, const uint32_t* __restrict__ alpha_full,
                           const uint32_t* __restrict__ kq, MsgB m, const uint32_t* __restrict__ g_w,
                           uint32_t* __restrict__ alpha, uint8_t* __restrict__ ok) {
  // 4 lanes per incoming MessageB: lanes 0..2 do  k_i B | c1 B | c2 B'  (variable base), then  alpha G | z G | z' G
  __shared__ JacSlots vs, fs;
  const int gid = blockIdx.x * 64 + threadIdx.x, rv = gid >> 2, sub = gid & 3, base = (int)threadIdx.x - sub;
  const int P1 = d.S - 1;
  const bool live = rv < d.B * d.S * P1 * 2;
  const int rvc = live ? rv : 0;
  const int v = rvc & 1, pp = rvc >> 1, jj = pp % P1, pi = pp / P1, i = pi % d.S, b = pi / d.S, ind = ind_of(i, jj);
  const int in = mbin_rv[rvc];
  const ec::U256 al = ec::sc_reduce(alpha_full + (size_t)rvc * 64, 64);
  const ec::Aff Bpk = ec::aff_load(m.pk + (size_t)in * 16), BTpk = ec::aff_load(m.tpk + (size_t)in * 16);
  const ec::Aff R1 = ec::aff_load(m.R + (size_t)in * 16), R2 = ec::aff_load(m.tR + (size_t)in * 16);
  ec::Jac res = ec::jac_inf(), fr = ec::jac_inf();
  if (live && sub < 3) {
    const ec::U256 sc = sub == 0 ? ec::u256_load(kq + (size_t)pi * 8) : (sub == 1 ? dlog_challenge(R1, Bpk) : dlog_challenge(R2, BTpk));
    res = ec::jac_mul(sc, sub == 2 ? BTpk : Bpk);
    const ec::U256 fk = sub == 0 ? al : ec::sc_reduce((sub == 1 ? m.z : m.tz) + (size_t)in * 8, 8);
    fr = ec::jac_mul_gen(fk);
  }
  vs.v[threadIdx.x] = res;
  fs.v[threadIdx.x] = fr;
  __syncthreads();
  if (!live || sub) return;
  ec::u256_store(alpha + (size_t)rv * 8, al);
  bool good = ec::jac_eq(fs.v[base], ec::jac_add_aff(vs.v[base], BTpk));                     // g^alpha == k_i B + B'
  good = good && ec::jac_eq_aff(ec::jac_add(fs.v[base + 1], vs.v[base + 1]), R1)             // DLogProof::verify x2
              && ec::jac_eq_aff(ec::jac_add(fs.v[base + 2], vs.v[base + 2]), R2);
  if (v == 1) good = good && ec::aff_eq(Bpk, ec::aff_load(g_w + (size_t)(b * d.S + ind) * 16));   // rounds.rs:281
  ok[rv] = good ? 1 : 0;
}

// G lanes per party (a power of two >= 2 S): lane 2j | 2j+1 does z1_j G | z2_j H, lane j also e_j T_j
__global__ void __launch_bounds__(64) r3_group_kernel(Dim d, int G, const uint32_t* __restrict__ delta_i, Ped p, uint32_t* __restrict__ dinv, uint8_t* __restrict__ ok) {
  __shared__ JacSlots vs, fs;
  const int gid = blockIdx.x * 64 + threadIdx.x, pi = gid / G, sub = gid % G, base = (int)threadIdx.x - sub;
  const bool live = pi < d.B * d.S;
  const int b = live ? pi / d.S : 0;
  const ec::Aff Gp = ec::aff_gen(), H = ec::aff_h2();
  ec::Jac res = ec::jac_inf(), fr = ec::jac_inf();
  if (live && sub < d.S) {
    const size_t o = (size_t)b * d.S + sub;
    const ec::Aff T = ec::aff_load(p.T + o * 16), a1 = ec::aff_load(p.a1 + o * 16), a2 = ec::aff_load(p.a2 + o * 16);
    const ec::Aff hp[5] = {Gp, H, T, a1, a2};
    res = ec::jac_mul(hash_points(hp), T);
  }
  if (live && sub < 2 * d.S) {
    const size_t o = (size_t)b * d.S + (sub >> 1);
    fr = ec::jac_mul_fixed(ec::u256_load(((sub & 1) ? p.z2 : p.z1) + o * 8), sub & 1);
  }
  vs.v[threadIdx.x] = res;
  fs.v[threadIdx.x] = fr;
  __syncthreads();
  if (!live || sub) return;
  ec::U256 sum = ec::u256_zero();
  bool good = true;
  for (int j = 0; j < d.S; ++j) {
    const size_t o = (size_t)b * d.S + j;
    sum = ec::sc_add(sum, ec::u256_load(delta_i + o * 8));
    const ec::Aff a1 = ec::aff_load(p.a1 + o * 16), a2 = ec::aff_load(p.a2 + o * 16);
    const ec::Jac lhs = ec::jac_add(fs.v[base + 2 * j], fs.v[base + 2 * j + 1]);
    const ec::Jac rhs = ec::jac_add_aff(ec::jac_add_aff(vs.v[base + j], a1), a2);
    good = good && ec::jac_eq(lhs, rhs);
  }
  good = good && !ec::u256_is_zero(sum);
  ec::u256_store(dinv + (size_t)pi * 8, ec::sc_inv(sum));
  ok[pi] = good ? 1 : 0;
}

// G lanes per party (a power of two >= 3 S): lane 3j+k does e D_j | z2_j R | e E_j; lane 2j+k does z1_j H | z2_j G
__global__ void __launch_bounds__(64) r6_group_kernel(Dim d, int G, const uint32_t* __restrict__ R, const uint32_t* __restrict__ pedT, Heg h,
                          const uint32_t* __restrict__ y, uint8_t* __restrict__ ok) {
  __shared__ JacSlots vs, fs;
  const int gid = blockIdx.x * 64 + threadIdx.x, pi = gid / G, sub = gid % G, base = (int)threadIdx.x - sub;
  const bool live = pi < d.B * d.S;
  const int pic = live ? pi : 0, b = pic / d.S;
  const ec::Aff Gp = ec::aff_gen(), H = ec::aff_h2(), Rp = ec::aff_load(R + (size_t)pic * 16);
  ec::Jac res = ec::jac_inf(), fr = ec::jac_inf();
  if (live && sub < 3 * d.S) {
    const int j = sub / 3, kind = sub % 3;
    const size_t o = (size_t)b * d.S + j;
    const ec::Aff TT = ec::aff_load(h.T + o * 16), A3 = ec::aff_load(h.A3 + o * 16), D = ec::aff_load(pedT + o * 16),
                  E = ec::aff_load(h.S + o * 16);
    const ec::Aff hp[7] = {TT, A3, Rp, H, Gp, D, E};
    const ec::U256 e = hash_points(hp);
    res = ec::jac_mul(kind == 1 ? ec::u256_load(h.z2 + o * 8) : e, kind == 0 ? D : (kind == 1 ? Rp : E));
  }
  if (live && sub < 2 * d.S) {
    const size_t o = (size_t)b * d.S + (sub >> 1);
    fr = ec::jac_mul_fixed(ec::u256_load(((sub & 1) ? h.z2 : h.z1) + o * 8), (sub & 1) ? 0 : 1);
  }
  vs.v[threadIdx.x] = res;
  fs.v[threadIdx.x] = fr;
  __syncthreads();
  if (!live || sub) return;
  bool good = true;
  ec::Jac acc = ec::jac_inf();
  for (int j = 0; j < d.S; ++j) {
    const size_t o = (size_t)b * d.S + j;
    const ec::Aff TT = ec::aff_load(h.T + o * 16), A3 = ec::aff_load(h.A3 + o * 16), E = ec::aff_load(h.S + o * 16);
    const ec::Jac l1 = ec::jac_add(fs.v[base + 2 * j], fs.v[base + 2 * j + 1]);
    const ec::Jac r1 = ec::jac_add_aff(vs.v[base + 3 * j], TT);
    const ec::Jac r2 = ec::jac_add_aff(vs.v[base + 3 * j + 2], A3);
    good = good && ec::jac_eq(l1, r1) && ec::jac_eq(vs.v[base + 3 * j + 1], r2);
    acc = ec::jac_add_aff(acc, E);
  }
  good = good && ec::jac_eq_aff(acc, ec::aff_load(y));
  ok[pi] = good ? 1 : 0;
}

// ---- Round 7: local signatures, output_signature, verify (party_i.rs:850-936) -------------------------------
struct Flags { const uint8_t *vi, *rv, *r3, *r4, *r5, *r6; };
__global__ void __launch_bounds__(64) r7_kernel(Dim d, Flags f, const uint32_t* __restrict__ msg, const uint32_t* __restrict__ R,
                          const uint32_t* __restrict__ kq, const uint32_t* __restrict__ sigma_i, const uint32_t* __restrict__ y,
                          uint32_t* __restrict__ r_out, uint32_t* __restrict__ s_out, int32_t* __restrict__ recid_out,
                          uint32_t* __restrict__ R_out, int32_t* __restrict__ status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int S = d.S, P1 = S - 1;
  int st = 0;
  const int nvi = S * P1 * d.V * d.n, nrv = S * P1 * 2;
  for (int q = 0; q < nvi && !st; ++q) if (!f.vi[(size_t)b * nvi + q]) st = 101;          // MessageB::b -> InvalidKey
  for (int q = 0; q < nrv && !st; ++q) if (!f.rv[(size_t)b * nrv + q]) st = 201;          // verify_proofs_get_alpha
  for (int j = 0; j < S && !st; ++j) if (!f.r3[(size_t)b * S + j]) st = 302;
  for (int j = 0; j < S && !st; ++j) if (!f.r4[(size_t)b * S + j]) st = 401;
  for (int j = 0; j < S && !st; ++j) if (!f.r5[(size_t)b * S + j]) st = 501;
  for (int j = 0; j < S && !st; ++j) if (!f.r6[(size_t)b * S + j]) st = 601;
  const ec::Aff Rp = ec::aff_load(R + (size_t)b * S * 16);
  const ec::U256 m = ec::sc_reduce(msg + (size_t)b * 8, 8), r = ec::sc_reduce(Rp.x.w, 8);
  ec::U256 s = ec::u256_zero();
  for (int j = 0; j < S; ++j) {
    const size_t o = (size_t)b * S + j;
    s = ec::sc_add(s, ec::sc_add(ec::sc_mul(m, ec::u256_load(kq + o * 8)), ec::sc_mul(r, ec::u256_load(sigma_i + o * 8))));
  }
  const ec::U256 ry = ec::sc_reduce(Rp.y.w, 8);
  int recid = (int)(ry.w[0] & 1u);
  const ec::U256 neg = ec::sc_neg(s);
  {  // if s > q - s: s = q - s, recid ^= 1
    bool gt = false;
    for (int j = 7; j >= 0; --j) { if (s.w[j] != neg.w[j]) { gt = s.w[j] > neg.w[j]; break; } }
    if (gt) { s = neg; recid ^= 1; }
  }
  // verify (party_i.rs:913-936)
  bool okv = !ec::u256_is_zero(s);
  if (okv) {
    const ec::U256 bi = ec::sc_inv(s), u1 = ec::sc_mul(m, bi), u2 = ec::sc_mul(r, bi);
    const ec::Aff V = ec::jac_to_aff(ec::jac_add(ec::jac_mul_gen(u1), ec::jac_mul(u2, ec::aff_load(y))));
    okv = !V.inf && ec::u256_eq(ec::sc_reduce(V.x.w, 8), r);
  }
  if (!okv && !st) st = 701;
  ec::u256_store(r_out + (size_t)b * 8, r);
  ec::u256_store(s_out + (size_t)b * 8, s);
  recid_out[b] = recid;
  if (R_out) ec::aff_store(R_out + (size_t)b * 16, Rp);
  status[b] = st;
}

template <class T>
static T* W(Seq& q, size_t count) {
  T* p = ws_array<T>(q.ctx, count);
  if (!p && q.rc == MPE_OK) { q.rc = MPE_E_NOMEM; mpe_set_error_msg("gg20: workspace under-reserved"); }
  return p;
}
// MPE_GG20_TRACE=1 in the environment: synchronise after every step and report it on stderr (debug aid)
static void gg_trace(hipStream_t st, const char* what, int rc) {
  static const bool on = getenv("MPE_GG20_TRACE") != nullptr;
  if (!on) return;
  const hipError_t e = hipStreamSynchronize(st);
  fprintf(stderr, "[gg20] %-28s rc=%d sync=%s\n", what, rc, hipGetErrorString(e));
  fflush(stderr);
}
#define GG_LAUNCH(kernel, nitems, ...)                                                                   \
  do {                                                                                                    \
    if (q.rc == MPE_OK && (nitems) > 0)                                                                   \
      hipLaunchKernelGGL(kernel, dim3(blocks_for((int)(nitems), 64)), dim3(64), 0, st, __VA_ARGS__);      \
    gg_trace(st, #kernel, q.rc);                                                                          \
  } while (0)

// one chunk of sessions [b0, b0+B)
static int sign_chunk(mpe_ctx* ctx, const mpe_gg20_keys* K, int B, int b0, const mpe_gg20_nonces* Z, uint32_t* d_r,
                      uint32_t* d_s, int32_t* d_recid, uint32_t* d_R, int32_t* d_status, int dedup, hipStream_t st) {
  const int S = K->S, n = K->n, P1 = S - 1, P = S * P1;
  Dim d{B, S, n, dedup ? 1 : 2, dedup ? 1 : S};
  const size_t nPI = (size_t)B * S, nAP = nPI * n, nVI = (size_t)B * P * d.V * n, nMB = (size_t)B * P * 2, nPP = (size_t)B * P,
               nPV = (size_t)B * d.PV * P;
  // workspace: own arrays + the largest inner composite (alice_verify over nVI items / pdl_verify over nPV items)
  const size_t own = nPI * 700 + nAP * 260 + nVI * 8 + nMB * 720 + nPP * 470 + nPV * 8 + 64 * 64;
  const size_t inner = (nVI > nPV ? nVI : nPV) * 2300 + nAP * 2100 + nMB * 300;
  // the inner composites call ws_reserve themselves (it resets the bump pointer), so this function keeps
  // its own arrays in a second arena carved from the tail of one reservation: reserve everything once here
  // and let the inner calls see a workspace that is already large enough (ws_reserve then only resets ws_off).
  MPE_TRY(ws_reserve(ctx, (own + inner) * 4 + (1u << 20), st));
  // carve the "own" region at the top of the workspace so that inner resets do not touch it
  char* top = (char*)ctx->ws + ctx->ws_bytes;
  size_t top_off = 0;
  auto own_alloc = [&](size_t bytes) -> void* {
    top_off = (top_off + bytes + 255) & ~(size_t)255;
    return top - top_off;
  };
  auto OW = [&](size_t words) { return (uint32_t*)own_alloc(words * 4); };
  auto OI = [&](size_t count) { return (int32_t*)own_alloc(count * 4); };
  auto OF = [&](size_t count) { return (uint8_t*)own_alloc(count); };
  Seq q{ctx, st, B};

  // nonce slices of this chunk
  const size_t oPI = (size_t)b0 * S, oAP = oPI * n, oMB = (size_t)b0 * P * 2, oPP = (size_t)b0 * P;
  const uint32_t *z_k = Z->k + oPI * 8, *z_gamma = Z->gamma + oPI * 8, *z_blind = Z->blind + oPI * 8, *z_ra = Z->r_a + oPI * 64;
  mpe_alice_nonces an{Z->al_alpha + oAP * 24, Z->al_beta + oAP * 64, Z->al_gamma + oAP * 88, Z->al_rho + oAP * 72};
  mpe_pdl_nonces pn{Z->pdl_alpha + oPP * 24, Z->pdl_beta + oPP * 64, Z->pdl_rho + oPP * 72, Z->pdl_gamma + oPP * 88};

  Idx ix;
  ix.key_pi = OI(nPI);
  ix.pi_ap = OI(nAP); ix.key_ap = OI(nAP); ix.st_ap = OI(nAP);
  ix.ap_vi = OI(nVI); ix.pia_vi = OI(nVI); ix.key_vi = OI(nVI); ix.st_vi = OI(nVI);
  ix.pia_mb = OI(nMB); ix.key_mb = OI(nMB); ix.mbin_rv = OI(nMB); ix.key_rv = OI(nMB);
  ix.pi_pp = OI(nPP); ix.key_pp = OI(nPP); ix.st_pp = OI(nPP);
  ix.pp_pv = OI(nPV); ix.pip_pv = OI(nPV); ix.key_pv = OI(nPV); ix.st_pv = OI(nPV);
  size_t total = nVI;
  if (nMB > total) total = nMB;
  if (nAP > total) total = nAP;
  if (nPV > total) total = nPV;
  if (nPI > total) total = nPI;
  GG_LAUNCH(idx_kernel, total, d, K->d_signers, ix, (int)total);

  // ---- Round 0 ----
  uint32_t *kq = OW(nPI * 8), *gq = OW(nPI * 8), *w = OW(nPI * 8), *k64 = OW(nPI * 64), *g_gamma = OW(nPI * 16),
           *g_w = OW(nPI * 16), *com = OW(nPI * 8), *c_a = OW(nPI * 128);
  GG_LAUNCH(r0_kernel, nPI, d, K->d_signers, K->x, K->X, z_k, z_gamma, z_blind, kq, gq, w, k64, g_gamma, g_w, com);
  if (q.rc == MPE_OK) q.rc = paillier_encrypt(ctx, K->pk, (int)nPI, ix.key_pi, k64, z_ra, c_a, true, st);          // MessageA.c
  gg_trace(st, "encrypt k", q.rc);
  mpe_alice_proof ap{OW(nAP * 64), OW(nAP * 8), OW(nAP * 64), OW(nAP * 25), OW(nAP * 89)};
  if (q.rc == MPE_OK)
    q.rc = alice_generate(ctx, K->pk, K->stm, (int)nAP, ix.key_ap, ix.st_ap, rows(kq, 8, ix.pi_ap), rows(c_a, 128, ix.pi_ap),
                          rows(z_ra, 64, ix.pi_ap), &an, &ap, st);
  gg_trace(st, "alice_generate", q.rc);

  // ---- Round 1 ----
  uint8_t* ok_vi = OF(nVI);
  if (q.rc == MPE_OK) {
    AliceProofRows pr{rows(ap.z, 64, ix.ap_vi), rows(ap.e, 8, ix.ap_vi), rows(ap.s, 64, ix.ap_vi), rows(ap.s1, 25, ix.ap_vi),
                      rows(ap.s2, 89, ix.ap_vi)};
    q.rc = alice_verify(ctx, K->pk, K->stm, (int)nVI, ix.key_vi, ix.st_vi, rows(c_a, 128, ix.pia_vi), pr, ok_vi, st);
  }
  gg_trace(st, "alice_verify", q.rc);
  uint32_t *bsel = OW(nMB * 8), *btq = OW(nMB * 8), *beta = OW(nMB * 8),
           *c_b = OW(nMB * 128);
  uint32_t *Bpk = OW(nMB * 16), *BR = OW(nMB * 16), *Bz = OW(nMB * 8), *BTpk = OW(nMB * 16), *BTR = OW(nMB * 16), *BTz = OW(nMB * 8);
  const uint32_t *z_bt = Z->mb_beta_tag + oMB * 64, *z_mr = Z->mb_r + oMB * 64, *z_nb = Z->mb_nonce_b + oMB * 8,
                 *z_nbt = Z->mb_nonce_bt + oMB * 8;
  GG_LAUNCH(mb_prep_kernel, nMB, d, gq, w, z_bt, bsel, btq, beta);
  if (q.rc == MPE_OK)                                                           // encrypt, Paillier::mul, Paillier::add :133-145
    q.rc = paillier_mul_add_enc(ctx, K->pk, (int)nMB, ix.key_mb, rows(c_a, 128, ix.pia_mb), rows(bsel, 8), 8, z_bt, z_mr, c_b, st);
  gg_trace(st, "MessageB ciphertext", q.rc);
  GG_LAUNCH(dlog_prove_kernel, nMB, (int)nMB, bsel, z_nb, Bpk, BR, Bz);                                       // :147
  GG_LAUNCH(dlog_prove_kernel, nMB, (int)nMB, btq, z_nbt, BTpk, BTR, BTz);                                    // :148

  // ---- Round 2 ----
  uint32_t *alpha_full = OW(nMB * 64), *alpha = OW(nMB * 8);
  uint8_t* ok_rv = OF(nMB);
  if (q.rc == MPE_OK) {
    // Paillier::decrypt of the incoming c_b with my key (mta/mod.rs:165); rows gathered receiver-ordered
    uint32_t* cin = OW(nMB * 128);
    GG_LAUNCH(gather_rows_kernel, nMB * 128, (int)nMB, 128, c_b, ix.mbin_rv, cin);
    q.rc = paillier_decrypt(ctx, K->pk, (int)nMB, ix.key_rv, cin, alpha_full, st);
  }
  gg_trace(st, "decrypt", q.rc);
  MsgB mbv{Bpk, BR, Bz, BTpk, BTR, BTz};
  const size_t lanes_fit = ctx->ec_lane_groups ? (size_t)ctx->cus * 4 * 64 * 2 : 0;      // two waves per SIMD
  if (nMB * 4 <= lanes_fit) GG_LAUNCH(r2a_group_kernel, nMB * 4, d, ix.mbin_rv, alpha_full, kq, mbv, g_w, alpha, ok_rv);
  else GG_LAUNCH(r2a_kernel, nMB, d, ix.mbin_rv, alpha_full, kq, mbv, g_w, alpha, ok_rv);
  uint32_t *delta_i = OW(nPI * 8), *sigma_i = OW(nPI * 8), *lq = OW(nPI * 8);
  Ped ped{OW(nPI * 16), OW(nPI * 16), OW(nPI * 16), OW(nPI * 8), OW(nPI * 8)};
  GG_LAUNCH(r2b_kernel, nPI, d, kq, gq, w, alpha, beta, Z->l + oPI * 8, Z->ped_s1 + oPI * 8, Z->ped_s2 + oPI * 8, delta_i,
            sigma_i, lq, ped);

  // ---- Round 3, 4 ----
  uint32_t *dinv = OW(nPI * 8), *R = OW(nPI * 16), *Rbar = OW(nPI * 16);
  uint8_t *ok_r3 = OF(nPI), *ok_r4 = OF(nPI), *ok_r5 = OF(nPI), *ok_r6 = OF(nPI);
  const int g3 = 2 * S <= 4 ? 4 : (2 * S <= 8 ? 8 : 16), g6 = 3 * S <= 8 ? 8 : (3 * S <= 16 ? 16 : 32);
  if (nPI * g3 <= lanes_fit) GG_LAUNCH(r3_group_kernel, nPI * g3, d, g3, delta_i, ped, dinv, ok_r3);
  else GG_LAUNCH(r3_kernel, nPI, d, delta_i, ped, dinv, ok_r3);
  GG_LAUNCH(r4_kernel, nPI, d, dinv, g_gamma, com, z_blind, Bpk, kq, R, Rbar, ok_r4);
  mpe_pdl_proof pp{OW(nPP * 64), OW(nPP * 16), OW(nPP * 128), OW(nPP * 64), OW(nPP * 25), OW(nPP * 64), OW(nPP * 89)};
  if (q.rc == MPE_OK)                                                                                          // phase5_proof_pdl
    q.rc = pdl_prove(ctx, K->pk, K->stm, (int)nPP, ix.key_pp, ix.st_pp, rows(c_a, 128, ix.pi_pp), rows(Rbar, 16, ix.pi_pp),
                     rows(R, 16, ix.pi_pp), rows(kq, 8, ix.pi_pp), rows(z_ra, 64, ix.pi_pp), &pn, &pp, st);
  gg_trace(st, "pdl_prove", q.rc);

  // ---- Round 5, 6, 7 ----
  uint8_t* ok_pv = OF(nPV);
  if (q.rc == MPE_OK) {
    PdlProofRows pr{rows(pp.z, 64, ix.pp_pv), rows(pp.u1, 16, ix.pp_pv), rows(pp.u2, 128, ix.pp_pv), rows(pp.u3, 64, ix.pp_pv),
                    rows(pp.s1, 25, ix.pp_pv), rows(pp.s2, 64, ix.pp_pv), rows(pp.s3, 89, ix.pp_pv)};
    q.rc = pdl_verify(ctx, K->pk, K->stm, (int)nPV, ix.key_pv, ix.st_pv, rows(c_a, 128, ix.pip_pv), rows(Rbar, 16, ix.pip_pv),
                      rows(R, 16, ix.pip_pv), pr, ok_pv, st);
  }
  gg_trace(st, "pdl_verify", q.rc);
  Heg heg{OW(nPI * 16), OW(nPI * 16), OW(nPI * 16), OW(nPI * 8), OW(nPI * 8)};
  GG_LAUNCH(r5_kernel, nPI, d, ok_pv, R, Rbar, sigma_i, lq, ped.T, Z->heg_s1 + oPI * 8, Z->heg_s2 + oPI * 8, heg, ok_r5);
  if (nPI * g6 <= lanes_fit) GG_LAUNCH(r6_group_kernel, nPI * g6, d, g6, R, ped.T, heg, K->y, ok_r6);
  else GG_LAUNCH(r6_kernel, nPI, d, R, ped.T, heg, K->y, ok_r6);
  Flags fl{ok_vi, ok_rv, ok_r3, ok_r4, ok_r5, ok_r6};
  GG_LAUNCH(r7_kernel, B, d, fl, Z->msg + (size_t)b0 * 8, R, kq, sigma_i, K->y, d_r + (size_t)b0 * 8, d_s + (size_t)b0 * 8,
            d_recid + b0, d_R ? d_R + (size_t)b0 * 16 : nullptr, d_status + b0);
  if (top_off + ctx->ws_off > ctx->ws_bytes && q.rc == MPE_OK) {
    q.rc = MPE_E_NOMEM;
    mpe_set_error_msg("gg20: workspace regions overlap (internal sizing error)");
  }
  return q.finish("gg20 sign_chunk");
}

}  // namespace gg
}  // namespace mpe

extern "C" {

int mpe_gg20_keys_create(mpe_ctx* ctx, int t, int n, int n_signers, const int32_t* h_signers, const uint32_t* d_x,
                         const uint32_t* d_p, const uint32_t* d_q, const uint32_t* d_Nt, const uint32_t* d_h1,
                         const uint32_t* d_h2, const uint32_t* d_y, const uint32_t* d_X, mpe_gg20_keys** out, void* stream) {
  if (!ctx || !h_signers || !d_x || !d_p || !d_q || !d_Nt || !d_h1 || !d_h2 || !d_y || !d_X || !out) return MPE_E_ARG;
  if (n < 2 || n > 8 || n_signers < 2 || n_signers > n || t < 1 || n_signers != t + 1) return MPE_E_ARG;
  for (int i = 0; i < n_signers; ++i)
    if (h_signers[i] < 0 || h_signers[i] >= n || (i && h_signers[i] <= h_signers[i - 1])) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  mpe_gg20_keys* K = new (std::nothrow) mpe_gg20_keys();
  if (!K) return MPE_E_NOMEM;
  K->t = t; K->n = n; K->S = n_signers;
  for (int i = 0; i < n_signers; ++i) K->signers[i] = h_signers[i];
  const size_t words = (size_t)n * 8 + (size_t)n * 16 + 16 + 8;
  hipError_t e = hipMalloc(&K->blob, words * 4);
  if (e != hipSuccess) { delete K; mpe_set_error("hipMalloc(gg20 keys)", e); return MPE_E_NOMEM; }
  K->x = (uint32_t*)K->blob; K->X = K->x + (size_t)n * 8; K->y = K->X + (size_t)n * 16; K->d_signers = (int32_t*)(K->y + 16);
  (void)hipMemcpyAsync(K->x, d_x, (size_t)n * 8 * 4, hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(K->X, d_X, (size_t)n * 16 * 4, hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(K->y, d_y, 16 * 4, hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(K->d_signers, K->signers, (size_t)n_signers * 4, hipMemcpyHostToDevice, st);
  int rc = mpe_paillier_create_private(ctx, n, d_p, d_q, &K->pk, stream);
  if (rc == MPE_OK) rc = mpe_statements_create(ctx, n, d_Nt, d_h1, d_h2, &K->stm, stream);
  if (rc != MPE_OK) { mpe_gg20_keys_destroy(K); return rc; }
  (void)hipStreamSynchronize(st);     // K->signers (host) was the source of an async copy
  *out = K;
  return MPE_OK;
}

int mpe_gg20_keys_destroy(mpe_gg20_keys* K) {
  if (!K) return MPE_E_ARG;
  if (K->pk) mpe_paillier_destroy(K->pk);
  if (K->stm) mpe_statements_destroy(K->stm);
  if (K->blob) (void)hipFree(K->blob);
  delete K;
  return MPE_OK;
}

int mpe_gg20_sign(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, const mpe_gg20_nonces* nonces, uint32_t* d_r,
                  uint32_t* d_s, int32_t* d_recid, uint32_t* d_R, int32_t* d_status, int dedup_verify, int chunk,
                  void* stream) {
  if (!ctx || !keys || !nonces || !d_r || !d_s || !d_recid || !d_status || batch < 0) return MPE_E_ARG;
  if (chunk <= 0) {
    // 65 536 sessions per pass (~13 GB of workspace at t=1, n=3; the EC kernels want >= 2 waves per SIMD), fewer
    // when the shape is wide: the workspace per session grows like S (S-1) n, keep a pass under ~64 GB
    const size_t S = keys->S, n = keys->n, P = S * (S - 1), V = dedup_verify ? 1 : 2, PV = dedup_verify ? 1 : S;
    const size_t nVI = P * V * n, nPV = PV * P;
    const size_t words = S * 700 + S * n * 260 + nVI * 8 + P * 2 * 720 + P * 470 + nPV * 8 +
                         (nVI > nPV ? nVI : nPV) * 2300 + S * n * 2100 + P * 2 * 300;          // per session, as sign_chunk sizes it
    size_t fit = ((size_t)64 << 30) / (words * 4);
    fit = fit >= 1024 ? (fit / 1024) * 1024 : (fit ? fit : 1);
    chunk = (int)(fit < 65536 ? fit : 65536);
  }
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int B = batch - b0 < chunk ? batch - b0 : chunk;
    int rc = mpe::gg::sign_chunk(ctx, keys, B, b0, nonces, d_r, d_s, d_recid, d_R, d_status, dedup_verify, (hipStream_t)stream);
    if (rc != MPE_OK) return rc;
  }
  return MPE_OK;
}

}  // extern "C"
