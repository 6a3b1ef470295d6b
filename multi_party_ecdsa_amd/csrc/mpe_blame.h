// Identifiable abort of GG20 signing, batched: src/protocols/multi_party_ecdsa/gg_2020/blame.rs
//   GlobalStatePhase5::phase5_blame   :116-224   (R_dash sum failed: somebody's delta_i / MtA messages are wrong)
//   GlobalStatePhase6::phase6_blame   :322-421   (S_i sum failed: somebody's sigma_i is wrong)
//   GlobalStatePhase7::phase7_blame   :434-454   (the signature failed: somebody's s_i is wrong)
// plus curv's ECDDHProof (blame.rs:258-272) as stand-alone entry points.  Every signer has OPENED the values the failing
// phase used; the functions re-derive the public ciphertexts from the openings (Paillier encryptions under the signers'
// PUBLIC keys: the same two-base ladders and range of kernels the signing rounds use) and name the parties whose openings do
// not match, or whose broadcast value is inconsistent.  Item layout: [B][S] (signer ordinal), then the peer slot j (S-1;
// ind = j < i ? j : j+1).  Output: one bit mask over signer ordinals per session — the reference's sorted, de-duplicated
// `bad_actors` (its Err is returned even when the mask is empty: the caller invokes blame because a check already failed).
// Included by mpe_lib.hip.
#pragma once
#include "mpe_gg20.h"

namespace mpe {
namespace bl {

using gg::Dim;
using gg::ind_of;

// item tables: pi = b*S + i, pp = pi*(S-1) + j
struct BIdx { int32_t *key_pi, *key_pp, *pi_pp, *gam_pp; };
__global__ void bidx_kernel(Dim d, BIdx x) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int S = d.S, P1 = S - 1;
  if (g < d.B * S) x.key_pi[g] = gg::kpub(d, g / S, g % S);
  if (g < d.B * S * P1) {
    const int j = g % P1, pi = g / P1, i = pi % S, b = pi / S;
    x.key_pp[g] = gg::kpub(d, b, i); x.pi_pp[g] = pi; x.gam_pp[g] = b * S + ind_of(i, j);
  }
}
__global__ void widen_kernel(int n, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {      // [n][8] -> [n][64]
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * 64) return;
  const size_t i = g / 64, w = g % 64;
  dst[g] = w < 8 ? src[i * 8 + w] : 0u;
}
// flag[i] = rows equal
__global__ void rows_eq_kernel(int n, int words, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint8_t* __restrict__ eq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t o = 0;
  for (int w = 0; w < words; ++w) o |= a[(size_t)i * words + w] ^ b[(size_t)i * words + w];
  eq[i] = o == 0;
}
// flag[pi] = (gamma_i G == g_gamma_i)      blame.rs:121-125
__global__ void __launch_bounds__(64) MPE_EC_OCC gamma_check_kernel(int n, const uint32_t* __restrict__ gamma, const uint32_t* __restrict__ g_gamma,
                                                         uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ok[i] = ec::jac_eq_aff(ec::jac_mul_gen(ec::sc_reduce(gamma + (size_t)i * 8, 8)), ec::aff_load(g_gamma + (size_t)i * 16)) ? 1 : 0;
}
// the sequential logic of phase5_blame over the flags, and the delta reconstruction (:127-211); one session per lane
__global__ void __launch_bounds__(64) MPE_EC_OCC blame5_combine_kernel(Dim d, const uint8_t* __restrict__ g_ok, const uint8_t* __restrict__ ca_ok,
                                                            const uint8_t* __restrict__ cb_ok, const uint32_t* __restrict__ k,
                                                            const uint32_t* __restrict__ gamma, const uint32_t* __restrict__ beta_tag,
                                                            const uint32_t* __restrict__ delta, uint32_t* __restrict__ bad_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int S = d.S, P1 = S - 1;
  uint32_t bad = 0;
  for (int i = 0; i < S; ++i) if (!g_ok[(size_t)b * S + i]) bad |= 1u << i;
  for (int i = 0; i < S; ++i) {
    if (!ca_ok[(size_t)b * S + i]) bad |= 1u << i;
    if (bad) continue;                                                               // :140 `if bad_signers_vec.is_empty()`
    for (int j = 0; j < P1; ++j) if (!cb_ok[((size_t)b * S + i) * P1 + j]) bad |= 1u << ind_of(i, j);
  }
  if (!bad) {
    for (int i = 0; i < S; ++i) {
      const ec::U256 ki = ec::sc_reduce(k + ((size_t)b * S + i) * 8, 8);
      ec::U256 acc = ec::sc_mul(ki, ec::sc_reduce(gamma + ((size_t)b * S + i) * 8, 8));
      for (int j = 0; j < P1; ++j) {
        // alpha_ij = k_i gamma_ind - beta_ij, beta_ij = -beta_tag_ij;  plus the beta of the MtA where i played Bob
        const int ind = ind_of(i, j), ind2 = j < i ? i - 1 : i;
        const ec::U256 bt = ec::sc_reduce(beta_tag + (((size_t)b * S + i) * P1 + j) * 64, 64);
        acc = ec::sc_add(acc, ec::sc_add(ec::sc_mul(ki, ec::sc_reduce(gamma + ((size_t)b * S + ind) * 8, 8)), bt));
        acc = ec::sc_sub(acc, ec::sc_reduce(beta_tag + (((size_t)b * S + ind) * P1 + ind2) * 64, 64));
      }
      if (!ec::u256_eq(acc, ec::sc_reduce(delta + ((size_t)b * S + i) * 8, 8))) bad |= 1u << i;
    }
  }
  bad_out[b] = bad;
}

// g_ni[pp] = k_i g_w[ind] - miu_ij G       blame.rs:360-376
__global__ void __launch_bounds__(64) MPE_EC_OCC gni_kernel(Dim d, const uint32_t* __restrict__ k, const uint32_t* __restrict__ miu, const uint32_t* __restrict__ gw,
                                                 uint32_t* __restrict__ gni) {
  const int pp = blockIdx.x * blockDim.x + threadIdx.x;
  const int S = d.S, P1 = S - 1;
  if (pp >= d.B * S * P1) return;
  const int j = pp % P1, pi = pp / P1, i = pi % S, b = pi / S, ind = ind_of(i, j);
  const ec::Jac a = ec::jac_mul(ec::sc_reduce(k + (size_t)pi * 8, 8), ec::aff_load(gw + ((size_t)gg::ks_of(d, b) * S + ind) * 16));
  const ec::Jac m = ec::jac_mul_gen(ec::sc_neg(ec::sc_reduce(miu + (size_t)pp * 64, 64)));
  ec::aff_store(gni + (size_t)pp * 16, ec::jac_to_aff(ec::jac_add(a, m)));
}
struct Ecddh { const uint32_t *a1, *a2, *z; };
__device__ inline bool ecddh_verify(const ec::Aff& g1, const ec::Aff& h1, const ec::Aff& g2, const ec::Aff& h2, const ec::Aff& a1,
                                    const ec::Aff& a2, const ec::U256& z, const ec::Enc& enc) {
  if (!(ec::aff_valid(g1) && ec::aff_valid(h1) && ec::aff_valid(g2) && ec::aff_valid(h2) && ec::aff_valid(a1) && ec::aff_valid(a2))) return false;
  const ec::Aff hp[6] = {g1, h1, g2, h2, a1, a2};
  const ec::U256 e = gg::hash_points(hp, enc, enc.ord_ecddh);
  return ec::jac_eq(ec::jac_mul(z, g1), ec::jac_add_aff(ec::jac_mul(e, h1), a1)) &&
         ec::jac_eq(ec::jac_mul(z, g2), ec::jac_add_aff(ec::jac_mul(e, h2), a2));
}
// g_sigma_i and the ECDDH proof of signer i  (:380-414)
__global__ void __launch_bounds__(64) MPE_EC_OCC gsigma_kernel(Dim d, const uint32_t* __restrict__ k, const uint32_t* __restrict__ miu, const uint32_t* __restrict__ gw,
                                                    const uint32_t* __restrict__ gni, const uint32_t* __restrict__ R, const uint32_t* __restrict__ Svec,
                                                    Ecddh pr, uint8_t* __restrict__ ok) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  const int S = d.S, P1 = S - 1;
  if (pi >= d.B * S) return;
  const int i = pi % S, b = pi / S;
  ec::Jac acc = ec::jac_mul(ec::sc_reduce(k + (size_t)pi * 8, 8), ec::aff_load(gw + ((size_t)gg::ks_of(d, b) * S + i) * 16));
  ec::U256 ms = ec::u256_zero();
  for (int j = 0; j < P1; ++j) ms = ec::sc_add(ms, ec::sc_reduce(miu + ((size_t)pi * P1 + j) * 64, 64));
  acc = ec::jac_add(acc, ec::jac_mul_gen(ms));
  for (int j = 0; j < P1; ++j) {
    const int ind1 = ind_of(i, j), ind2 = j < i ? i - 1 : i;
    acc = ec::jac_add_aff(acc, ec::aff_load(gni + (((size_t)b * S + ind1) * P1 + ind2) * 16));
  }
  const ec::Aff gs = ec::jac_to_aff(acc);
  ok[pi] = ecddh_verify(ec::aff_gen(), gs, ec::aff_load(R + (size_t)b * 16), ec::aff_load(Svec + (size_t)pi * 16),
                        ec::aff_load(pr.a1 + (size_t)pi * 16), ec::aff_load(pr.a2 + (size_t)pi * 16), ec::sc_reduce(pr.z + (size_t)pi * 8, 8), d.enc) ? 1 : 0;
}
__global__ void blame6_combine_kernel(Dim d, const uint8_t* __restrict__ mu_ok, const uint8_t* __restrict__ ca_ok, const uint8_t* __restrict__ dd_ok,
                                      uint32_t* __restrict__ bad_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int S = d.S, P1 = S - 1;
  uint32_t bad = 0;
  for (int i = 0; i < S; ++i) {
    for (int j = 0; j < P1; ++j) if (!mu_ok[((size_t)b * S + i) * P1 + j]) bad |= 1u << i;
    if (!ca_ok[(size_t)b * S + i]) bad |= 1u << i;
  }
  if (!bad) for (int i = 0; i < S; ++i) if (!dd_ok[(size_t)b * S + i]) bad |= 1u << i;
  bad_out[b] = bad;
}
// phase7_blame: R s_i == m R_dash_i + r S_i   (:434-454)
__global__ void __launch_bounds__(64) MPE_EC_OCC blame7_kernel(int B, int S, const uint32_t* __restrict__ s, const uint32_t* __restrict__ r, const uint32_t* __restrict__ Rdash,
                                                    const uint32_t* __restrict__ m, const uint32_t* __restrict__ R, const uint32_t* __restrict__ Svec,
                                                    uint8_t* __restrict__ ok) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= B * S) return;
  const int b = pi / S;
  const ec::Aff Rp = ec::aff_load(R + (size_t)b * 16), Rd = ec::aff_load(Rdash + (size_t)pi * 16), Sp = ec::aff_load(Svec + (size_t)pi * 16);
  if (!(ec::aff_valid(Rp) && ec::aff_valid(Rd) && ec::aff_valid(Sp))) { ok[pi] = 0; return; }
  const ec::Jac l = ec::jac_mul(ec::sc_reduce(s + (size_t)pi * 8, 8), Rp);
  const ec::Jac rr = ec::jac_add(ec::jac_mul(ec::sc_reduce(m + (size_t)b * 8, 8), Rd), ec::jac_mul(ec::sc_reduce(r + (size_t)b * 8, 8), Sp));
  ok[pi] = ec::jac_eq(l, rr) ? 1 : 0;
}
__global__ void mask_from_flags_kernel(int B, int S, const uint8_t* __restrict__ ok, uint32_t* __restrict__ bad_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  uint32_t bad = 0;
  for (int i = 0; i < S; ++i) if (!ok[(size_t)b * S + i]) bad |= 1u << i;
  bad_out[b] = bad;
}
__global__ void __launch_bounds__(64) MPE_EC_OCC ecddh_prove_kernel(int B, ec::Enc enc, const uint32_t* __restrict__ x, const uint32_t* __restrict__ s_in, const uint32_t* __restrict__ g1,
                                                         const uint32_t* __restrict__ h1, const uint32_t* __restrict__ g2, const uint32_t* __restrict__ h2,
                                                         uint32_t* __restrict__ a1, uint32_t* __restrict__ a2, uint32_t* __restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 xx = ec::sc_reduce(x + (size_t)i * 8, 8), s = ec::sc_reduce(s_in + (size_t)i * 8, 8);
  const ec::Aff G1 = ec::aff_load(g1 + (size_t)i * 16), H1 = ec::aff_load(h1 + (size_t)i * 16), G2 = ec::aff_load(g2 + (size_t)i * 16),
                H2 = ec::aff_load(h2 + (size_t)i * 16);
  const ec::Aff A1 = gg::mul_aff(s, G1), A2 = gg::mul_aff(s, G2);
  const ec::Aff hp[6] = {G1, H1, G2, H2, A1, A2};
  const ec::U256 e = gg::hash_points(hp, enc, enc.ord_ecddh);
  ec::aff_store(a1 + (size_t)i * 16, A1);
  ec::aff_store(a2 + (size_t)i * 16, A2);
  ec::u256_store(z + (size_t)i * 8, ec::sc_add(s, ec::sc_mul(e, xx)));
}
__global__ void __launch_bounds__(64) MPE_EC_OCC ecddh_verify_kernel(int B, ec::Enc enc, const uint32_t* __restrict__ g1, const uint32_t* __restrict__ h1, const uint32_t* __restrict__ g2,
                                                          const uint32_t* __restrict__ h2, Ecddh pr, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  ok[i] = ecddh_verify(ec::aff_load(g1 + (size_t)i * 16), ec::aff_load(h1 + (size_t)i * 16), ec::aff_load(g2 + (size_t)i * 16),
                       ec::aff_load(h2 + (size_t)i * 16), ec::aff_load(pr.a1 + (size_t)i * 16), ec::aff_load(pr.a2 + (size_t)i * 16),
                       ec::sc_reduce(pr.z + (size_t)i * 8, 8), enc) ? 1 : 0;
}
// the session's openings for phase-6 blame: the ECDDH proof that S_i = sigma_i R (blame.rs:258-272), sigma_i never leaves
__global__ void __launch_bounds__(64) MPE_EC_OCC session_ecddh_kernel(Dim d, const uint32_t* __restrict__ sigma_i, const uint32_t* __restrict__ R,
                                                           const uint32_t* __restrict__ nonce, uint32_t* __restrict__ a1, uint32_t* __restrict__ a2,
                                                           uint32_t* __restrict__ z) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const size_t o = (size_t)(pi % d.L) * d.B + pi / d.L;
  const ec::U256 x = ec::u256_load(sigma_i + (size_t)pi * 8), s = ec::sc_reduce(nonce + (size_t)pi * 8, 8);
  const ec::Aff G1 = ec::aff_gen(), G2 = ec::aff_load(R + (size_t)pi * 16);
  const ec::Aff H1 = ec::jac_to_aff(ec::jac_mul_gen(x)), H2 = gg::mul_aff(x, G2);
  const ec::Aff A1 = ec::jac_to_aff(ec::jac_mul_gen(s)), A2 = gg::mul_aff(s, G2);
  const ec::Aff hp[6] = {G1, H1, G2, H2, A1, A2};
  const ec::U256 e = gg::hash_points(hp, d.enc, d.enc.ord_ecddh);
  ec::aff_store(a1 + o * 16, A1);
  ec::aff_store(a2 + o * 16, A2);
  ec::u256_store(z + o * 8, ec::sc_add(s, ec::sc_mul(e, x)));
}
// [pi][S-1][64] -> [L][B][S-1][64]
__global__ void miu_out_kernel(Dim d, const uint32_t* __restrict__ miu, uint32_t* __restrict__ out) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)(d.S - 1) * 64;
  if (g >= (size_t)d.B * d.L * per) return;
  const size_t pi = g / per, w = g % per;
  out[((pi % d.L) * d.B + pi / d.L) * per + w] = miu[g];
}

static Dim blame_dim(const mpe_ctx* ctx, const mpe_gg20_keys* K, int B, const int32_t* d_keyset) {
  Dim d{};
  d.enc = ctx->enc;
  d.B = B; d.S = K->S; d.n = K->n; d.L = K->S; d.K = K->K; d.n_own = K->n_own; d.ks = d_keyset;
  for (int i = 0; i < 8; ++i) { d.loc[i] = i; d.sg[i] = K->signers[i]; d.oslot[i] = K->own_slot[i] < 0 ? 0 : K->own_slot[i]; }
  return d;
}

}  // namespace bl
}  // namespace mpe

// ---- kzen-paillier `Open` (blame.rs:252-256 extract_paillier_randomness) -------------------------------------------
// c = (1 + m N) r^N mod N^2  =>  r^N = c (mod N)  =>  r = (c mod N)^d mod N with d = N^-1 mod phi(N).
// phi is even: with u = phi^-1 mod N (N odd: the batched inversion) phi u = 1 + k N, so N (phi - k) = 1 (mod phi):
// d = phi - (phi u - 1) / N, the exact quotient through N^-1 mod 2^2048.
namespace mpe {
namespace bl {
__global__ void open_phi_kernel(int nk, const uint32_t* __restrict__ N, const uint32_t* __restrict__ pq32, uint32_t* __restrict__ phi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nk) return;
  uint32_t s[64];
  const uint32_t one[1] = {1u};
  sm::add(s, 64, pq32 + (size_t)(2 * k) * 32, 32, pq32 + (size_t)(2 * k + 1) * 32, 32);
  sm::sub(s, 64, s, 64, one, 1);                                           // p + q - 1
  sm::sub(phi + (size_t)k * 64, 64, N + (size_t)k * 64, 64, s, 64);        // phi = N - (p + q - 1)
}
__global__ void open_d_kernel(int nk, const uint32_t* __restrict__ N, const uint32_t* __restrict__ phi, const uint32_t* __restrict__ u,
                              const uint8_t* __restrict__ ok, uint32_t* __restrict__ d) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nk) return;
  uint32_t t[128], ninv[64], t1[64], t2[64];
  const uint32_t one[1] = {1u};
  sm::mul(t, phi + (size_t)k * 64, 64, u + (size_t)k * 64, 64);
  sm::sub(t, 128, t, 128, one, 1);                                         // phi u - 1 = k N
  sm::inv2adic(ninv, N + (size_t)k * 64, 64, t1, t2);
  sm::mullo(t1, t, ninv, 64);                                              // k (< phi < 2^2048)
  sm::sub(d + (size_t)k * 64, 64, phi + (size_t)k * 64, 64, t1, 64);
  if (!ok[k]) sm::zero(d + (size_t)k * 64, 64);                            // gcd(N, phi) != 1: not a Paillier key
}
}  // namespace bl
}  // namespace mpe

extern "C" {

int mpe_paillier_open(mpe_ctx* ctx, const mpe_paillier* sk, int batch, const int32_t* d_key_idx, const uint32_t* d_c, uint32_t* d_m,
                      uint32_t* d_r, void* stream) {
  using namespace mpe;
  if (!ctx || !sk || !d_c || !d_m || !d_r || batch < 0) return MPE_E_ARG;
  if (!sk->has_private) { mpe_set_error_msg("mpe_paillier_open: key set has no private part"); return MPE_E_ARG; }
  if (!d_key_idx && sk->nkeys != 1 && sk->nkeys < batch) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  MPE_TRY(paillier_decrypt(ctx, sk, batch, d_key_idx, rows(d_c, 128), d_m, st));
  const int nk = sk->nkeys;
  MPE_TRY(ws_reserve(ctx, ((size_t)nk * 64 * 3 + modinv_ws_words(sk->ms_n, nk)) * 4 + 65536, st));
  uint32_t *phi = ws_array<uint32_t>(ctx, (size_t)nk * 64), *u = ws_array<uint32_t>(ctx, (size_t)nk * 64), *d = ws_array<uint32_t>(ctx, (size_t)nk * 64);
  uint8_t* ok = ws_array<uint8_t>(ctx, ((size_t)nk + 255) & ~(size_t)255);
  if (!phi || !u || !d || !ok) { mpe_set_error_msg("mpe_paillier_open: workspace"); return MPE_E_NOMEM; }
  MPE_LAUNCH_1D(bl::open_phi_kernel, nk, st, nk, sk->N, sk->pq32, phi);
  MPE_TRY(launch_modinv(ctx, sk->ms_n, nk, rows(nullptr, 1), rows(phi, 64), u, ok, st));
  MPE_LAUNCH_1D(bl::open_d_kernel, nk, st, nk, sk->N, phi, u, ok, d);
  // r = (c mod N)^d mod N: the ciphertext enters the 2048-bit engine as a double-width base
  const int rc = launch_modexp(ctx, sk->ms_n, batch, key_selector(sk, d_key_idx), Rows{d_c, nullptr, 128, 64}, Rows{d_c + 64, nullptr, 128, 64},
                               key_rows(sk, d, 64, d_key_idx), 64, d_r, st);
  (void)hipMemsetAsync(d, 0, (size_t)nk * 64 * 4, st);                     // d is as secret as p, q
  (void)hipMemsetAsync(phi, 0, (size_t)nk * 64 * 4, st);
  (void)hipMemsetAsync(u, 0, (size_t)nk * 64 * 4, st);
  return rc;
}

int mpe_gg20_blame5(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, const int32_t* d_keyset, const mpe_gg20_blame5_in* in,
                    uint32_t* d_bad_actors, void* stream) {
  if (!ctx || !keys || !in || !d_bad_actors || batch < 0 || (keys->K > 1 && !d_keyset)) return MPE_E_ARG;
  if (!in->k || !in->k_rand || !in->gamma || !in->beta_tag || !in->beta_rand || !in->delta || !in->g_gamma || !in->c_a || !in->c_b) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  using namespace mpe;
  hipStream_t st = (hipStream_t)stream;
  const int S = keys->S, P1 = S - 1, nPI = batch * S, nPP = nPI * P1;
  const bl::Dim d = bl::blame_dim(ctx, keys, batch, d_keyset);
  // own arrays at the top of the workspace (never handed out, never moved: ws_top), the Paillier composites below
  const size_t own = ((size_t)nPI * (64 + 128) + (size_t)nPP * 128) * 4 + ((size_t)nPI + 3 * (size_t)nPP) * 4 + (size_t)nPI * 2 + nPP + 16 * 256;   // BIdx: one [nPI] + three [nPP] arrays; 256 B slack per take()
  MPE_TRY(ws_reserve(ctx, own + ws_need_encrypt(nPI) + ws_need_mul_add_enc(nPP) + (1u << 20), st));
  char* top = (char*)ctx->ws + ctx->ws_bytes;
  auto take = [&](size_t bytes) { top -= (bytes + 255) & ~(size_t)255; return (void*)top; };
  bl::BIdx ix{(int32_t*)take((size_t)nPI * 4), (int32_t*)take((size_t)nPP * 4), (int32_t*)take((size_t)nPP * 4), (int32_t*)take((size_t)nPP * 4)};
  uint32_t *k64 = (uint32_t*)take((size_t)nPI * 64 * 4), *ca = (uint32_t*)take((size_t)nPI * 128 * 4), *cb = (uint32_t*)take((size_t)nPP * 128 * 4);
  uint8_t *g_ok = (uint8_t*)take(nPI), *ca_ok = (uint8_t*)take(nPI), *cb_ok = (uint8_t*)take(nPP);
  ctx->ws_top = (size_t)(((char*)ctx->ws + ctx->ws_bytes) - top);
  int rc = MPE_OK;
  hipLaunchKernelGGL(bl::bidx_kernel, dim3(blocks_for(nPP > nPI ? nPP : nPI, 64)), dim3(64), 0, st, d, ix);
  hipLaunchKernelGGL(bl::gamma_check_kernel, dim3(blocks_for(nPI, 64)), dim3(64), 0, st, nPI, in->gamma, in->g_gamma, g_ok);
  hipLaunchKernelGGL(bl::widen_kernel, dim3(blocks_for(nPI * 64, 256)), dim3(256), 0, st, nPI, in->k, k64);
  // MessageA::a_with_predefined_randomness(k_i, ek_i, k_randomness_i, &[]).c  == m_a_vec[i].c          :128-138
  rc = paillier_encrypt(ctx, keys->pub, nPI, ix.key_pi, k64, in->k_rand, ca, false, st);
  if (rc == MPE_OK) {
    hipLaunchKernelGGL(bl::rows_eq_kernel, dim3(blocks_for(nPI, 64)), dim3(64), 0, st, nPI, 128, ca, in->c_a, ca_ok);
    // MessageB::b_with_predefined_randomness(gamma_ind, ek_i, message_a, beta_randomness, beta_tag, &[]).c == m_b_mat[i][j].c   :144-156
    rc = paillier_mul_add_enc(ctx, keys->pub, nPP, ix.key_pp, rows(ca, 128, ix.pi_pp), rows(in->gamma, 8, ix.gam_pp), 8, in->beta_tag, in->beta_rand, cb, st);
  }
  if (rc == MPE_OK) {
    hipLaunchKernelGGL(bl::rows_eq_kernel, dim3(blocks_for(nPP, 64)), dim3(64), 0, st, nPP, 128, cb, in->c_b, cb_ok);
    hipLaunchKernelGGL(bl::blame5_combine_kernel, dim3(blocks_for(batch, 64)), dim3(64), 0, st, d, g_ok, ca_ok, cb_ok, in->k, in->gamma, in->beta_tag,
                       in->delta, d_bad_actors);
  }
  ctx->ws_top = 0;
  if (rc != MPE_OK) return rc;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("mpe_gg20_blame5", e); return MPE_E_HIP; }
  return MPE_OK;
}

int mpe_gg20_blame6(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, const int32_t* d_keyset, const mpe_gg20_blame6_in* in,
                    uint32_t* d_bad_actors, void* stream) {
  if (!ctx || !keys || !in || !d_bad_actors || batch < 0 || (keys->K > 1 && !d_keyset)) return MPE_E_ARG;
  if (!in->k || !in->k_rand || !in->miu || !in->miu_rand || !in->a1 || !in->a2 || !in->z || !in->S || !in->c_a || !in->c_b || !in->R) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  using namespace mpe;
  hipStream_t st = (hipStream_t)stream;
  const int S = keys->S, P1 = S - 1, nPI = batch * S, nPP = nPI * P1;
  const bl::Dim d = bl::blame_dim(ctx, keys, batch, d_keyset);
  const size_t own = ((size_t)nPI * (64 + 128) + (size_t)nPP * (128 + 16)) * 4 + ((size_t)nPI + 3 * (size_t)nPP) * 4 + (size_t)nPI * 2 + nPP + 16 * 256;   // BIdx: one [nPI] + three [nPP] arrays; 256 B slack per take()
  MPE_TRY(ws_reserve(ctx, own + ws_need_encrypt(nPP) + (1u << 20), st));
  char* top = (char*)ctx->ws + ctx->ws_bytes;
  auto take = [&](size_t bytes) { top -= (bytes + 255) & ~(size_t)255; return (void*)top; };
  bl::BIdx ix{(int32_t*)take((size_t)nPI * 4), (int32_t*)take((size_t)nPP * 4), (int32_t*)take((size_t)nPP * 4), (int32_t*)take((size_t)nPP * 4)};
  uint32_t *k64 = (uint32_t*)take((size_t)nPI * 64 * 4), *ca = (uint32_t*)take((size_t)nPI * 128 * 4), *cb = (uint32_t*)take((size_t)nPP * 128 * 4),
           *gni = (uint32_t*)take((size_t)nPP * 16 * 4);
  uint8_t *dd_ok = (uint8_t*)take(nPI), *ca_ok = (uint8_t*)take(nPI), *mu_ok = (uint8_t*)take(nPP);
  ctx->ws_top = (size_t)(((char*)ctx->ws + ctx->ws_bytes) - top);
  hipLaunchKernelGGL(bl::bidx_kernel, dim3(blocks_for(nPP > nPI ? nPP : nPI, 64)), dim3(64), 0, st, d, ix);
  hipLaunchKernelGGL(bl::widen_kernel, dim3(blocks_for(nPI * 64, 256)), dim3(256), 0, st, nPI, in->k, k64);
  // Enc_ek_i(miu_ij; miu_randomness_ij) == m_b_mat[i][j].c   :327-339     and     Enc(k_i) == m_a_vec[i].c   :342-354
  int rc = paillier_encrypt(ctx, keys->pub, nPP, ix.key_pp, in->miu, in->miu_rand, cb, false, st);
  if (rc == MPE_OK) {
    hipLaunchKernelGGL(bl::rows_eq_kernel, dim3(blocks_for(nPP, 64)), dim3(64), 0, st, nPP, 128, cb, in->c_b, mu_ok);
    rc = paillier_encrypt(ctx, keys->pub, nPI, ix.key_pi, k64, in->k_rand, ca, false, st);
  }
  if (rc == MPE_OK) {
    hipLaunchKernelGGL(bl::rows_eq_kernel, dim3(blocks_for(nPI, 64)), dim3(64), 0, st, nPI, 128, ca, in->c_a, ca_ok);
    hipLaunchKernelGGL(bl::gni_kernel, dim3(blocks_for(nPP, 64)), dim3(64), 0, st, d, in->k, in->miu, keys->gw, gni);
    hipLaunchKernelGGL(bl::gsigma_kernel, dim3(blocks_for(nPI, 64)), dim3(64), 0, st, d, in->k, in->miu, keys->gw, gni, in->R, in->S,
                       bl::Ecddh{in->a1, in->a2, in->z}, dd_ok);
    hipLaunchKernelGGL(bl::blame6_combine_kernel, dim3(blocks_for(batch, 64)), dim3(64), 0, st, d, mu_ok, ca_ok, dd_ok, d_bad_actors);
  }
  ctx->ws_top = 0;
  if (rc != MPE_OK) return rc;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("mpe_gg20_blame6", e); return MPE_E_HIP; }
  return MPE_OK;
}

int mpe_gg20_blame7(mpe_ctx* ctx, int n_signers, int batch, const mpe_gg20_blame7_in* in, uint32_t* d_bad_actors, void* stream) {
  if (!ctx || !in || !d_bad_actors || batch < 0 || n_signers < 2 || n_signers > 8) return MPE_E_ARG;
  if (!in->s || !in->r || !in->R_dash || !in->m || !in->R || !in->S) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  using namespace mpe;
  hipStream_t st = (hipStream_t)stream;
  const int nPI = batch * n_signers;
  MPE_TRY(ws_reserve(ctx, (size_t)nPI + 4096, st));
  uint8_t* ok = ws_array<uint8_t>(ctx, nPI);
  if (!ok) return MPE_E_NOMEM;
  hipLaunchKernelGGL(bl::blame7_kernel, dim3(blocks_for(nPI, 64)), dim3(64), 0, st, batch, n_signers, in->s, in->r, in->R_dash, in->m, in->R, in->S, ok);
  hipLaunchKernelGGL(bl::mask_from_flags_kernel, dim3(blocks_for(batch, 64)), dim3(64), 0, st, batch, n_signers, ok, d_bad_actors);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("mpe_gg20_blame7", e); return MPE_E_HIP; }
  return MPE_OK;
}

int mpe_ecddh_prove(mpe_ctx* ctx, int batch, const uint32_t* d_x, const uint32_t* d_s, const mpe_ecddh_statement* statement,
                    const mpe_ecddh_proof* out, void* stream) {
  if (!ctx || !d_x || !d_s || !statement || !out || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::bl::ecddh_prove_kernel, batch, st, batch, ctx->enc, d_x, d_s, statement->g1, statement->h1, statement->g2, statement->h2, out->a1, out->a2, out->z);
  return MPE_OK;
}
int mpe_ecddh_verify(mpe_ctx* ctx, int batch, const mpe_ecddh_statement* statement, const mpe_ecddh_proof* proof, uint8_t* d_ok, void* stream) {
  if (!ctx || !statement || !proof || !d_ok || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::bl::ecddh_verify_kernel, batch, st, batch, ctx->enc, statement->g1, statement->h1, statement->g2, statement->h2,
                mpe::bl::Ecddh{proof->a1, proof->a2, proof->z}, d_ok);
  return MPE_OK;
}

// What a party publishes for the phase-6 blame (LocalStatePhase6, blame.rs:227-234) beyond the values it was given as inputs
// (k_i, its MessageA randomness): miu [L][B][S-1][64] = the plaintexts of the w_i MtA before reduction, and the ECDDH proof that
// S_i = sigma_i R (nonce [B][L][8] an input like every other sampled value).  Valid after round 5.
int mpe_gg20_session_blame6_state(const mpe_gg20_session* s, const uint32_t* d_nonce, uint32_t* d_miu, uint32_t* d_a1, uint32_t* d_a2,
                                  uint32_t* d_z, void* stream) {
  if (!s || !d_nonce || !d_miu || !d_a1 || !d_a2 || !d_z) return MPE_E_ARG;
  if (s->next_round < 6) { mpe_set_error_msg("gg20: the phase-6 openings exist after round 5"); return MPE_E_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const mpe::gg::Counts c = mpe::gg::counts_of(s->d);
  hipLaunchKernelGGL(mpe::bl::miu_out_kernel, dim3(mpe::blocks_for((int)(c.nPP * 64), 256)), dim3(256), 0, st, s->d, s->miu, d_miu);
  hipLaunchKernelGGL(mpe::bl::session_ecddh_kernel, dim3(mpe::blocks_for((int)c.nPI, 64)), dim3(64), 0, st, s->d, s->sigma_i, s->R, d_nonce, d_a1, d_a2, d_z);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("mpe_gg20_session_blame6_state", e); return MPE_E_HIP; }
  return MPE_OK;
}

}  // extern "C"
