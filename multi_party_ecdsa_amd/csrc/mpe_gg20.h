// GG20 signing as the reference structures it — Round0..Round7 of
// src/protocols/multi_party_ecdsa/gg_2020/state_machine/sign/rounds.rs:68,122,234,347,431,525,612,672, each a function
// of (the party's state, the messages of the previous round) -> (next state, outgoing message) — batched over B
// sessions and over the LOCAL parties of those sessions:
//   * local = every signer: the whole session runs on this GPU in lock-step (what `round_based::dev::Simulation` does
//     for one session in the reference's own test, state_machine/sign.rs:667-763); mpe_gg20_sign is this composition;
//   * local = one signer: the per-party view a real deployment (or the party-sharded multi-GPU mode) uses — the
//     object only ever sees that party's secrets (x_i, p_i, q_i) and the broadcast messages.
// Party math: SignKeys / LocalSignature (src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:526-936), MessageA /
// MessageB (src/utilities/mta/mod.rs:52-179).  Every round is a handful of batched launches over
// (session, local party[, peer[, statement]]) items; messages are fixed-size records (include/mpecdsa_hip.h), read IN
// PLACE from the incoming slab through Rows views and packed into the outgoing one.
// Included by mpe_lib.hip.
//
// Index conventions (identical to oracle/gg20_oracle.c):
//   pi = b*L + li                     party instance (session b, local slot li; signer ordinal i = local[li])
//   ap = pi*n + st                    Alice range proof of pi for statement st
//   pp = pi*(S-1) + jj                ordered pair (pi -> peer slot jj), ind = jj < i ? jj : jj+1 (rounds.rs:149)
//   mb = pp*2 + v                     MessageB of pi for that peer, v = 0 (gamma_i) / 1 (w_i)
#pragma once
#include <cstdio>
#include <cstdlib>

#include "mpe_proofs.h"

namespace mpe {
namespace smp {
// the bounds of the sampling ranges of a key object (mpe_sample.h): q, q^3 and per party N - 2, q N~, q^3 N~
// (range_proofs.rs:48-51: q^3, q^3 N~, q N~;  zk_pdl_with_slack/mod.rs:69-77: the same and sample_range(1, N - 1) = 1 + below(N - 2))
struct Bounds { uint32_t *q, *q3, *Nm2, *qNt, *q3Nt; };
__global__ void bounds_kernel(int count, const uint32_t* __restrict__ N, const uint32_t* __restrict__ Nt, Bounds b) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > count) return;
  const uint32_t q[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  uint32_t q2[16], q3[24];
  sm::mul(q2, q, 8, q, 8);
  sm::mul(q3, q2, 16, q, 8);
  if (g == count) {                      // the two constants
    for (int j = 0; j < 8; ++j) b.q[j] = q[j];
    for (int j = 0; j < 24; ++j) b.q3[j] = q3[j];
    return;
  }
  const uint32_t two[1] = {2u};
  sm::sub(b.Nm2 + (size_t)g * 64, 64, N + (size_t)g * 64, 64, two, 1);
  sm::mul(b.qNt + (size_t)g * 72, Nt + (size_t)g * 64, 64, q, 8);
  sm::mul(b.q3Nt + (size_t)g * 88, Nt + (size_t)g * 64, 64, q3, 24);
}
}  // namespace smp
}  // namespace mpe

struct mpe_gg20_keys {
  int t = 0, n = 0, S = 0, K = 1, n_own = 0;
  int signers[8] = {0};
  int own[8] = {0};                 // party indices whose secrets this object holds
  int own_slot[8] = {0};            // party index -> slot in own, or -1
  mpe_paillier* pub = nullptr;      // K*n public Paillier keys (key set kk, party a = key kk*n + a)
  mpe_paillier* prv = nullptr;      // K*n_own private keys (kk*n_own + slot)
  mpe_statements* stm = nullptr;    // K*n statements
  void* blob = nullptr;
  size_t blob_bytes = 0;
  uint32_t* x = nullptr;            // [K][n_own][8]  key shares of the own parties
  uint32_t* X = nullptr;            // [K][n][16]     pk_vec
  uint32_t* y = nullptr;            // [K][16]        group public key
  uint32_t* gw = nullptr;           // [K][S][16]     g_w_vec = lambda_j X_j (SignKeys::g_w_vec, party_i.rs:527-544)
  mpe::smp::Bounds bounds{};        // sampling ranges (mpe_sample.h): q [8], q^3 [24], N-2 [K*n][64], q N~ [K*n][72], q^3 N~ [K*n][88]
};

namespace mpe {
namespace gg {

constexpr int SUB0 = 256, SUB1 = 208, W2 = 96, W3 = 24, SUB4 = 450, W5 = 64, W6 = 8;
inline int msg_words(int S, int n, int round) {
  switch (round) {
    case 0: return SUB0 * (n + 1);
    case 1: return SUB1 * 2 * (S - 1);
    case 2: return W2;
    case 3: return W3;
    case 4: return SUB4 * S;
    case 5: return W5;
    case 7: return W6;
    default: return 0;
  }
}

// V: range-proof verifications per MessageB pair (2 faithful / 1 dedup); PV: verifiers of the PDL proofs (L / 1)
struct Dim {
  int B, S, n, L, V, PV, K, n_own;
  int loc[8];        // local slot -> signer ordinal
  int sg[8];         // signer ordinal -> party index
  int oslot[8];      // party index -> slot among the own parties
  const int32_t* ks; // [B] key set of a session, or null
  ec::Enc enc;       // the context's transcript conventions (include/mpecdsa_hip.h: mpe_encoding), copied when the object is built
};
__device__ __forceinline__ int ind_of(int i, int jj) { return jj < i ? jj : jj + 1; }
__device__ __forceinline__ int jme_of(int i, int ind) { return i < ind ? i : i - 1; }
__device__ __forceinline__ int ks_of(const Dim& d, int b) { return d.ks ? d.ks[b] : 0; }
__device__ __forceinline__ int kpub(const Dim& d, int b, int i) { return ks_of(d, b) * d.n + d.sg[i]; }
__device__ __forceinline__ int kown(const Dim& d, int b, int i) { return ks_of(d, b) * d.n_own + d.oslot[d.sg[i]]; }

// incoming message slab: sender j's [B][W] block starts at record off[j]
struct Slab { const uint32_t* p; long long off[8]; int W; };
__device__ __forceinline__ long long rec_index(const Slab& s, int j, int b) { return s.off[j] + b; }
__device__ __forceinline__ const uint32_t* rec_of(const Slab& s, int j, int b) { return s.p + (size_t)(s.off[j] + b) * (size_t)s.W; }

// ---- index tables that do not depend on the message slabs (built once per session object) ------------------------
struct Idx {
  int32_t *kown_pi;                                  // [B*L]
  int32_t *pi_ap, *kown_ap, *st_ap;                  // [B*L*n]
  int32_t *ca_vi, *kpub_vi, *st_vi;                  // [B*L*(S-1)*V*n]  verifier-side range-proof checks
  int32_t *ca_mb, *kpub_mb, *kown_mb;                // [B*L*(S-1)*2]
  int32_t *pi_pp, *kown_pp, *st_pp;                  // [B*L*(S-1)]
  int32_t *ca_pv, *pi_pv, *kpub_pv, *st_pv;          // [B*PV*S*(S-1)]   PDL verifications
};
__global__ void idx_kernel(Dim d, Idx x, int total) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const int S = d.S, n = d.n, L = d.L, P1 = S - 1;
  if (g < d.B * L) { const int b = g / L; x.kown_pi[g] = kown(d, b, d.loc[g % L]); }
  if (g < d.B * L * n) {
    const int pi = g / n, b = pi / L;
    x.pi_ap[g] = pi; x.kown_ap[g] = kown(d, b, d.loc[pi % L]); x.st_ap[g] = ks_of(d, b) * n + g % n;
  }
  if (g < d.B * L * P1 * d.V * n) {
    const int st = g % n, r1 = g / n, r2 = r1 / d.V, jj = r2 % P1, pi = r2 / P1, i = d.loc[pi % L], b = pi / L;
    const int ind = ind_of(i, jj);
    x.ca_vi[g] = ind * d.B + b; x.kpub_vi[g] = kpub(d, b, ind); x.st_vi[g] = ks_of(d, b) * n + st;
  }
  if (g < d.B * L * P1 * 2) {
    const int pp = g >> 1, jj = pp % P1, pi = pp / P1, i = d.loc[pi % L], b = pi / L, ind = ind_of(i, jj);
    x.ca_mb[g] = ind * d.B + b; x.kpub_mb[g] = kpub(d, b, ind); x.kown_mb[g] = kown(d, b, i);
  }
  if (g < d.B * L * P1) {
    const int jj = g % P1, pi = g / P1, i = d.loc[pi % L], b = pi / L;
    x.pi_pp[g] = pi; x.kown_pp[g] = kown(d, b, i); x.st_pp[g] = ks_of(d, b) * n + d.sg[ind_of(i, jj)];
  }
  if (g < d.B * d.PV * S * P1) {
    const int jj = g % P1, r1 = g / P1, i = r1 % S, r2 = r1 / S, vl = r2 % d.PV, b = r2 / d.PV;
    x.ca_pv[g] = i * d.B + b; x.pi_pv[g] = b * L + vl; x.kpub_pv[g] = kpub(d, b, i);
    x.st_pv[g] = ks_of(d, b) * n + d.sg[ind_of(i, jj)];
  }
}

// dst[(j*B + b)*words + w] = record(j, b)[src_off + w]: keeps a field of every sender's message (m_a_vec, bc_vec, t_vec)
__global__ void gather_field_kernel(Slab s, int S, int B, int src_off, int words, uint32_t* __restrict__ dst) {
  MPE_FOREGROUND();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)S * B * words) return;
  const int w = (int)(g % words);
  const size_t jb = g / words;
  dst[g] = rec_of(s, (int)(jb / B), (int)(jb % B))[src_off + w];
}
// status / bad_actors of a party instance.  Every round has its own [pi] arrays (status + round * nPI): inside a round the
// FIRST failed check sticks, and the party's status is the first non-zero round — so a check that is evaluated late (the
// lock-step composition takes the pure verifications of rounds 1 and 5 off the critical path of small batches) still lands
// where the reference's order of evaluation puts it.
constexpr int NR = 9;              // rounds 0..7 and SignManual::complete
__device__ __forceinline__ void fail(int32_t* status, uint32_t* bad, int pi, int code, uint32_t mask) {
  if (status[pi] == 0) { status[pi] = code; bad[pi] = mask; }
}
__device__ __forceinline__ int resolve_status(const int32_t* status, size_t nPI, int pi, int upto, const uint32_t* bad, uint32_t* bad_out) {
  for (int r = 0; r <= upto; ++r) {
    const int st = status[(size_t)r * nPI + pi];
    if (st) { if (bad_out) *bad_out = bad[(size_t)r * nPI + pi]; return st; }
  }
  if (bad_out) *bad_out = 0;
  return 0;
}

// out record of (li, b), sub-record sub0 + (item % per):  [dst_off .. dst_off+words) = src[item].
// A party whose status is non-zero has stopped (RoundN::proceed returned Err / panicked in the reference): it sends zeros.
__global__ void pack_field_kernel(int nitems, int per, int L, int B, int nsub, int sub0, int subw, int dst_off,
                                  const uint32_t* __restrict__ src, int words, const int32_t* __restrict__ status, int round,
                                  uint32_t* __restrict__ out) {
  MPE_FOREGROUND();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)nitems * words) return;
  const int w = (int)(g % words), item = (int)(g / words), pi = item / per, li = pi % L, b = pi / L;
  const bool stopped = resolve_status(status, (size_t)L * B, pi, round, nullptr, nullptr) != 0;
  out[((size_t)(li * B + b) * nsub + sub0 + item % per) * subw + dst_off + w] = stopped ? 0u : src[g];
}
// Malformed points in the messages a party is about to read (off the curve, non-canonical coordinates, infinity): in the
// reference such a message does not even deserialise (curv's Point), so the round never sees it.  Here the party's status
// becomes 100*round + 90 with bad_actors = the senders, BEFORE any secret scalar is multiplied into such a point.
__device__ __forceinline__ bool pt_ok(const uint32_t* p) { return ec::aff_valid(ec::aff_load(p)); }
__global__ void __launch_bounds__(64) validate_kernel(Dim d, Slab in, int round, int32_t* __restrict__ status, uint32_t* __restrict__ bad) {
  MPE_FOREGROUND();
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int i = d.loc[pi % d.L], b = pi / d.L, P1 = d.S - 1;
  uint32_t mask = 0;
  for (int j = 0; j < d.S; ++j) {
    if (j == i) continue;
    const uint32_t* m = rec_of(in, j, b);
    bool good = true;
    if (round == 2) {
      for (int v = 0; v < 2; ++v) {
        const uint32_t* mb = m + (size_t)(jme_of(i, j) * 2 + v) * SUB1;
        good = good && pt_ok(mb + 128) && pt_ok(mb + 144) && pt_ok(mb + 168) && pt_ok(mb + 184);
      }
    } else if (round == 3) {
      good = pt_ok(m + 8) && pt_ok(m + 32) && pt_ok(m + 48) && pt_ok(m + 64);
    } else if (round == 4) {
      good = pt_ok(m + 8);
    } else if (round == 5) {
      for (int jj = 0; jj < P1; ++jj) good = good && pt_ok(m + (size_t)jj * SUB4 + 64);
      good = good && pt_ok(m + (size_t)P1 * SUB4);
    } else if (round == 6) {
      good = pt_ok(m) && pt_ok(m + 16) && pt_ok(m + 32);
    }
    if (!good) mask |= 1u << j;
  }
  if (mask) fail(status, bad, pi, 100 * round + 90, mask);
}

// ---- helpers -------------------------------------------------------------------------------------
__device__ inline ec::U256 lagrange0(const int* sg, int S, int i) {
  ec::U256 num = ec::u256_one(), den = ec::u256_one();
  for (int j = 0; j < S; ++j) {
    if (j == i) continue;
    ec::U256 xj = ec::u256_zero(); xj.w[0] = (uint32_t)(sg[j] + 1);
    ec::U256 xi = ec::u256_zero(); xi.w[0] = (uint32_t)(sg[i] + 1);
    num = ec::sc_mul(num, xj);
    den = ec::sc_mul(den, ec::sc_sub(xj, xi));
  }
  return ec::sc_mul(num, ec::sc_inv(den));
}
// HashCommitment(compressed point as BigInt, blind)  (party_i.rs:577-580)
__device__ inline ec::U256 commit_point(const ec::Aff& P, const uint32_t* blind, const ec::Enc& enc) {
  ec::Sha256 s; ec::sha_init(s);
  ec::sha_point_compressed(s, P);
  ec::sha_bigint(s, blind, 8, enc);
  return ec::sha_final(s);
}
// (the points are passed as an array reference and the function is force-inlined: handing a pointer to a
//  lane-private array to an out-of-line function hung the kernel on gfx950 / ROCm 7.2)
// `pts` is the canonical list of the proof (include/mpecdsa_hip.h names it per proof); `ord[i]` = which of them is hashed
// i-th.  The selection is a compare-and-select chain, never a dynamic index into the lane-private array.
template <int N>
__device__ __forceinline__ ec::U256 hash_points(const ec::Aff (&pts)[N], const ec::Enc& enc, const uint8_t* ord) {
  ec::Sha256 s; ec::sha_init(s);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int o = ord[i];
    ec::Aff P = pts[0];
#pragma unroll
    for (int j = 1; j < N; ++j) if (o == j) P = pts[j];
    ec::sha_chain_point(s, P, enc);
  }
  const ec::U256 d = ec::sha_final(s);
  return ec::sc_reduce(d.w, 8);
}
__device__ __forceinline__ ec::Aff mul_aff(const ec::U256& k, const ec::Aff& P) { return ec::jac_to_aff(ec::jac_mul(k, P)); }

// g_w_vec of a key set: lambda_j X_j for every signer (once per key object)
__global__ void __launch_bounds__(64) MPE_EC_OCC gw_kernel(Dim d, const uint32_t* __restrict__ Xs, uint32_t* __restrict__ gw) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= d.K * d.S) return;
  const int kk = g / d.S, j = g % d.S;
  ec::aff_store(gw + (size_t)g * 16, mul_aff(lagrange0(d.sg, d.S, j), ec::aff_load(Xs + ((size_t)kk * d.n + d.sg[j]) * 16)));
}

// ---- Round 0: SignKeys::create + phase1_broadcast (party_i.rs:546-589) ----------------------------
__global__ void __launch_bounds__(64) MPE_EC_OCC r0_kernel(Dim d, const uint32_t* __restrict__ xs, const uint32_t* __restrict__ k_in,
                          const uint32_t* __restrict__ gamma_in, const uint32_t* __restrict__ blind, uint32_t* __restrict__ kq,
                          uint32_t* __restrict__ gq, uint32_t* __restrict__ w, uint32_t* __restrict__ k64, uint32_t* __restrict__ g_gamma,
                          uint32_t* __restrict__ com, int32_t* __restrict__ status, uint32_t* __restrict__ bad) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int i = d.loc[pi % d.L], b = pi / d.L;
  // k_i, gamma_i are `Scalar::random()` (party_i.rs:561-563): non-zero and below q.  Anything else is not a value the reference can
  // hold — it is what the device sampler leaves when a rejection loop gave up (mpe_sample.h), or a caller's mistake — and the
  // party stops here: status MPE_GG20_STATUS_BAD_NONCE, no message leaves it
  const bool k_ok = ec::sc_is_canonical_nonzero(k_in + (size_t)pi * 8), g_ok = ec::sc_is_canonical_nonzero(gamma_in + (size_t)pi * 8);
  if (!k_ok || !g_ok) fail(status, bad, pi, MPE_GG20_STATUS_BAD_NONCE, 0);
  const ec::U256 k = ec::sc_reduce(k_in + (size_t)pi * 8, 8), g = ec::sc_reduce(gamma_in + (size_t)pi * 8, 8);
  const ec::U256 wi = ec::sc_mul(lagrange0(d.sg, d.S, i), ec::sc_reduce(xs + (size_t)kown(d, b, i) * 8, 8));
  ec::u256_store(kq + (size_t)pi * 8, k);
  ec::u256_store(gq + (size_t)pi * 8, g);
  ec::u256_store(w + (size_t)pi * 8, wi);
  for (int j = 0; j < 64; ++j) k64[(size_t)pi * 64 + j] = j < 8 ? k.w[j] : 0u;
  const ec::Aff gg = ec::jac_to_aff(ec::jac_mul_gen(g));
  ec::aff_store(g_gamma + (size_t)pi * 16, gg);
  ec::u256_store(com + (size_t)pi * 8, commit_point(gg, blind + (size_t)pi * 8, d.enc));
}

// ---- Round 1 -------------------------------------------------------------------------------------------------------
// sub-record index (in the incoming M0 slab) of the range proof every verification item reads
__global__ void idx1_kernel(Dim d, Slab in0, int32_t* __restrict__ sub0_vi) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int P1 = d.S - 1;
  if (g >= d.B * d.L * P1 * d.V * d.n) return;
  const int st = g % d.n, r1 = g / d.n, r2 = r1 / d.V, jj = r2 % P1, pi = r2 / P1, i = d.loc[pi % d.L], b = pi / d.L;
  sub0_vi[g] = (int32_t)(rec_index(in0, ind_of(i, jj), b) * (d.n + 1) + st);
}
// per MessageB the multiplier b, beta_tag mod q, beta = -beta_tag (mta/mod.rs:132,146)
__global__ void mb_prep_kernel(Dim d, const uint32_t* __restrict__ gq, const uint32_t* __restrict__ w,
                               const uint32_t* __restrict__ beta_tag, uint32_t* __restrict__ bsel,
                               uint32_t* __restrict__ btq, uint32_t* __restrict__ beta) {
  const int mb = blockIdx.x * blockDim.x + threadIdx.x;
  if (mb >= d.B * d.L * (d.S - 1) * 2) return;
  const int pi = (mb >> 1) / (d.S - 1);
  const uint32_t* src = (mb & 1) ? w + (size_t)pi * 8 : gq + (size_t)pi * 8;
  for (int j = 0; j < 8; ++j) bsel[(size_t)mb * 8 + j] = src[j];
  const ec::U256 t = ec::sc_reduce(beta_tag + (size_t)mb * 64, 64);
  ec::u256_store(btq + (size_t)mb * 8, t);
  ec::u256_store(beta + (size_t)mb * 8, ec::sc_neg(t));
}
// MessageB::b -> Err(InvalidKey) when a range proof of the peer fails (rounds.rs:151-175 -> Error::Round1)
__global__ void status1_kernel(Dim d, const uint8_t* __restrict__ ok_vi, int32_t* __restrict__ status, uint32_t* __restrict__ bad) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int per = (d.S - 1) * d.V * d.n;
  bool good = true;
  for (int q = 0; q < per; ++q) good = good && ok_vi[(size_t)pi * per + q];
  if (!good) fail(status, bad, pi, 101, 0);
}

// ---- Round 2 -------------------------------------------------------------------------------------------------------
// receiver view: sub-record (in the incoming M1 slab) of the MessageB that peer `ind` built for me
__global__ void idx2_kernel(Dim d, Slab in1, int32_t* __restrict__ sub1_rv) {
  MPE_FOREGROUND();
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int P1 = d.S - 1;
  if (g >= d.B * d.L * P1 * 2) return;
  const int v = g & 1, pp = g >> 1, jj = pp % P1, pi = pp / P1, i = d.loc[pi % d.L], b = pi / d.L, ind = ind_of(i, jj);
  sub1_rv[g] = (int32_t)(rec_index(in1, ind, b) * (2 * P1) + jme_of(i, ind) * 2 + v);
}
// MessageB::verify_proofs_get_alpha after the decryption (mta/mod.rs:166-178) and the check of rounds.rs:281
// code[rv]: 0 ok, 1 -> 201, 2 -> 202
__device__ __forceinline__ void r2a_finish(const Dim& d, int rv, int v, int pp, int b, int ind, const ec::Aff& Bpk, bool good,
                                           const uint32_t* gw, uint32_t* bpk_in, uint8_t* code) {
  uint8_t c = good ? 0 : 1;
  if (v == 1) { if (!c && !ec::aff_eq(Bpk, ec::aff_load(gw + ((size_t)ks_of(d, b) * d.S + ind) * 16))) c = 2; }   // rounds.rs:281
  else ec::aff_store(bpk_in + (size_t)pp * 16, Bpk);                                          // mb_gamma_s[jj].b_proof.pk, for phase4
  code[rv] = c;
}
__global__ void __launch_bounds__(64) MPE_EC_OCC r2a_kernel(Dim d, const int32_t* __restrict__ sub1_rv, const uint32_t* __restrict__ in1,
                           const uint32_t* __restrict__ alpha_full, const uint32_t* __restrict__ kq, const uint32_t* __restrict__ gw,
                           uint32_t* __restrict__ alpha, uint32_t* __restrict__ bpk_in, uint8_t* __restrict__ code) {
  MPE_FOREGROUND();
  const int rv = blockIdx.x * blockDim.x + threadIdx.x;
  const int P1 = d.S - 1;
  if (rv >= d.B * d.L * P1 * 2) return;
  const int v = rv & 1, pp = rv >> 1, jj = pp % P1, pi = pp / P1, i = d.loc[pi % d.L], b = pi / d.L, ind = ind_of(i, jj);
  const uint32_t* m = in1 + (size_t)sub1_rv[rv] * SUB1;
  const ec::U256 al = ec::sc_reduce(alpha_full + (size_t)rv * 64, 64);
  ec::u256_store(alpha + (size_t)rv * 8, al);
  const ec::Aff Bpk = ec::aff_load(m + 128), BTpk = ec::aff_load(m + 168);
  const ec::Jac g_alpha = ec::jac_mul_gen(al);
  const ec::Jac ba_btag = ec::jac_add_aff(ec::jac_mul(ec::u256_load(kq + (size_t)pi * 8), Bpk), BTpk);
  bool good = ec::jac_eq(g_alpha, ba_btag);
  {  // DLogProof::verify x2
    const ec::Aff R1 = ec::aff_load(m + 144), R2 = ec::aff_load(m + 184);
    const ec::U256 c1 = dlog_challenge(R1, Bpk, d.enc), c2 = dlog_challenge(R2, BTpk, d.enc);
    const ec::Jac l1 = ec::jac_add(ec::jac_mul_gen(ec::sc_reduce(m + 160, 8)), ec::jac_mul(c1, Bpk));
    const ec::Jac l2 = ec::jac_add(ec::jac_mul_gen(ec::sc_reduce(m + 200, 8)), ec::jac_mul(c2, BTpk));
    good = good && ec::jac_eq_aff(l1, R1) && ec::jac_eq_aff(l2, R2);
  }
  r2a_finish(d, rv, v, pp, b, ind, Bpk, good, gw, bpk_in, code);
}

// the three fixed-base multiplications of Round 2's T_i and PedersenProof::prove that need the party's nonces only (l H | s1 G | s2 H,
// affine, 48 words per party): Round 0 runs them beside its ladders, off the EC chain a small batch waits for between rounds 2 and 4
__global__ void __launch_bounds__(64) MPE_EC_OCC ped_ahead_kernel(Dim d, const uint32_t* __restrict__ l_in, const uint32_t* __restrict__ s1_in,
                                                       const uint32_t* __restrict__ s2_in, uint32_t* __restrict__ ped_pre) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  uint32_t* o = ped_pre + (size_t)pi * 48;
  ec::aff_store(o, ec::jac_to_aff(ec::jac_mul_h2(ec::sc_reduce(l_in + (size_t)pi * 8, 8))));
  ec::aff_store(o + 16, ec::jac_to_aff(ec::jac_mul_gen(ec::sc_reduce(s1_in + (size_t)pi * 8, 8))));
  ec::aff_store(o + 32, ec::jac_to_aff(ec::jac_mul_h2(ec::sc_reduce(s2_in + (size_t)pi * 8, 8))));
}
// delta_i, sigma_i, T_i + PedersenProof::prove (party_i.rs:591-634); the party's status of this round
struct Ped { uint32_t *T, *e, *a1, *a2, *z1, *z2; };       // [pi]
__global__ void __launch_bounds__(64) MPE_EC_OCC r2b_kernel(Dim d, const uint32_t* __restrict__ kq, const uint32_t* __restrict__ gq,
                           const uint32_t* __restrict__ w, const uint32_t* __restrict__ alpha,
                           const uint32_t* __restrict__ beta, const uint8_t* __restrict__ code, const uint32_t* __restrict__ l_in,
                           const uint32_t* __restrict__ s1_in, const uint32_t* __restrict__ s2_in,
                           uint32_t* __restrict__ delta_i, uint32_t* __restrict__ sigma_i, uint32_t* __restrict__ lq, Ped p,
                           int32_t* __restrict__ status, uint32_t* __restrict__ bad, const uint32_t* __restrict__ alpha_full,
                           uint32_t* __restrict__ miu, int fault_step, uint32_t fault_mask, const uint32_t* __restrict__ ped_pre) {
  MPE_FOREGROUND();
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int P1 = d.S - 1;
  for (int jj = 0; jj < P1; ++jj)                                                              // LocalStatePhase6::miu
    for (int w_ = 0; w_ < 64; ++w_) miu[((size_t)pi * P1 + jj) * 64 + w_] = alpha_full[(((size_t)pi * P1 + jj) * 2 + 1) * 64 + w_];
  const ec::U256 k = ec::u256_load(kq + (size_t)pi * 8);
  ec::U256 de = ec::sc_mul(k, ec::u256_load(gq + (size_t)pi * 8)), si = ec::sc_mul(k, ec::u256_load(w + (size_t)pi * 8));
  int st = 0;
  for (int jj = 0; jj < P1; ++jj) {
    const size_t m0 = ((size_t)pi * P1 + jj) * 2;
    de = ec::sc_add(de, ec::sc_add(ec::u256_load(alpha + m0 * 8), ec::u256_load(beta + m0 * 8)));
    si = ec::sc_add(si, ec::sc_add(ec::u256_load(alpha + (m0 + 1) * 8), ec::u256_load(beta + (m0 + 1) * 8)));
    for (int v = 0; v < 2 && !st; ++v) if (code[m0 + v]) st = 200 + code[m0 + v];          // the loop order of rounds.rs:260-286
  }
  if (st) fail(status, bad, pi, st, 0);
  if ((fault_mask >> d.loc[pi % d.L]) & 1u) {          // the corrupt_step of the reference's tests (test.rs:458-465)
    if (fault_step == 5) de = ec::sc_add(de, de);
    if (fault_step == 6) si = ec::sc_add(si, si);
  }
  ec::u256_store(delta_i + (size_t)pi * 8, de);
  ec::u256_store(sigma_i + (size_t)pi * 8, si);
  const ec::U256 l = ec::sc_reduce(l_in + (size_t)pi * 8, 8);
  ec::u256_store(lq + (size_t)pi * 8, l);
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2();
  // l H, a1 = s1 G, a2 = s2 H depend on the party's own nonces only: Round 0 computed them beside its ladders (ped_ahead_kernel)
  const uint32_t* pre = ped_pre ? ped_pre + (size_t)pi * 48 : nullptr;
  const ec::Aff T = ec::jac_to_aff(ec::jac_add(ec::jac_mul_gen(si), pre ? ec::jac_from_aff(ec::aff_load(pre)) : ec::jac_mul_h2(l)));
  const ec::U256 s1 = ec::sc_reduce(s1_in + (size_t)pi * 8, 8), s2 = ec::sc_reduce(s2_in + (size_t)pi * 8, 8);
  const ec::Aff a1 = pre ? ec::aff_load(pre + 16) : ec::jac_to_aff(ec::jac_mul_gen(s1)), a2 = pre ? ec::aff_load(pre + 32) : ec::jac_to_aff(ec::jac_mul_h2(s2));
  const ec::Aff hp[5] = {G, H, T, a1, a2};
  const ec::U256 e = hash_points(hp, d.enc, d.enc.ord_pedersen);
  ec::aff_store(p.T + (size_t)pi * 16, T);
  ec::u256_store(p.e + (size_t)pi * 8, e);
  ec::aff_store(p.a1 + (size_t)pi * 16, a1);
  ec::aff_store(p.a2 + (size_t)pi * 16, a2);
  ec::u256_store(p.z1 + (size_t)pi * 8, ec::sc_add(s1, ec::sc_mul(e, si)));
  ec::u256_store(p.z2 + (size_t)pi * 8, ec::sc_add(s2, ec::sc_mul(e, l)));
}

// ---- Round 3: T_i == proof.com, delta^-1, every PedersenProof (rounds.rs:347-402) -----------------------------------
// M2 record: delta 0 | T 8 | e 24 | a1 32 | a2 48 | com 64 | z1 80 | z2 88
__device__ __forceinline__ bool words_eq(const uint32_t* a, const uint32_t* b, int n) {
  uint32_t o = 0;
  for (int i = 0; i < n; ++i) o |= a[i] ^ b[i];
  return o == 0;
}
__device__ __forceinline__ void r3_finish(int pi, bool com_ok, bool ped_ok, const ec::U256& sum, uint32_t* dinv, int32_t* status,
                                          uint32_t* bad) {
  if (!com_ok) fail(status, bad, pi, 303, 0);
  const bool inv_ok = !ec::u256_is_zero(sum);
  if (!inv_ok) fail(status, bad, pi, 301, 0);
  ec::u256_store(dinv + (size_t)pi * 8, inv_ok ? ec::sc_inv(sum) : ec::u256_zero());
  if (!ped_ok) fail(status, bad, pi, 302, 0);
}
__global__ void __launch_bounds__(64) MPE_EC_OCC r3_kernel(Dim d, Slab in2, uint32_t* __restrict__ dinv, int32_t* __restrict__ status,
                                                uint32_t* __restrict__ bad) {
  MPE_FOREGROUND();
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int b = pi / d.L;
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2();
  ec::U256 sum = ec::u256_zero();
  bool com_ok = true, ped_ok = true;
  for (int j = 0; j < d.S; ++j) {
    const uint32_t* m = rec_of(in2, j, b);
    sum = ec::sc_add(sum, ec::sc_reduce(m, 8));
    com_ok = com_ok && words_eq(m + 8, m + 64, 16);
    const ec::Aff C = ec::aff_load(m + 64), a1 = ec::aff_load(m + 32), a2 = ec::aff_load(m + 48);
    const ec::Aff hp[5] = {G, H, C, a1, a2};
    const ec::U256 e = hash_points(hp, d.enc, d.enc.ord_pedersen);
    const ec::Jac lhs = ec::jac_add(ec::jac_mul_gen(ec::sc_reduce(m + 80, 8)), ec::jac_mul_h2(ec::sc_reduce(m + 88, 8)));
    const ec::Jac rhs = ec::jac_add_aff(ec::jac_add_aff(ec::jac_mul(e, C), a1), a2);
    ped_ok = ped_ok && ec::jac_eq(lhs, rhs);
  }
  r3_finish(pi, com_ok, ped_ok, sum, dinv, status, bad);
}

// ---- Round 4: phase4 -> R, R_dash (party_i.rs:642-687, rounds.rs:452) --------------------------------
// M3 record: blind 0 | g_gamma 8
__global__ void __launch_bounds__(64) MPE_EC_OCC r4_kernel(Dim d, Slab in3, const uint32_t* __restrict__ dinv, const uint32_t* __restrict__ com_all,
                          const uint32_t* __restrict__ bpk_in, const uint32_t* __restrict__ kq, uint32_t* __restrict__ R,
                          uint32_t* __restrict__ Rbar, int32_t* __restrict__ status, uint32_t* __restrict__ bad, int with_rbar) {
  MPE_FOREGROUND();
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int P1 = d.S - 1, i = d.loc[pi % d.L], b = pi / d.L;
  uint32_t mask = 0;
  for (int jj = 0; jj < P1; ++jj) {
    const int ind = ind_of(i, jj);
    const uint32_t* m = rec_of(in3, ind, b);
    const ec::Aff gg = ec::aff_load(m + 8);
    bool good = ec::aff_eq(ec::aff_load(bpk_in + ((size_t)pi * P1 + jj) * 16), gg);
    good = good && ec::u256_eq(commit_point(gg, m, d.enc), ec::u256_load(com_all + ((size_t)ind * d.B + b) * 8));
    if (!good) mask |= 1u << ind;
  }
  if (mask) fail(status, bad, pi, 401, mask);
  ec::Jac acc = ec::jac_inf();
  for (int j = 0; j < d.S; ++j) acc = ec::jac_add(acc, ec::jac_from_aff(ec::aff_load(rec_of(in3, j, b) + 8)));
  const ec::Aff Rp = mul_aff(ec::u256_load(dinv + (size_t)pi * 8), ec::jac_to_aff(acc));
  ec::aff_store(R + (size_t)pi * 16, Rp);
  if (with_rbar) ec::aff_store(Rbar + (size_t)pi * 16, mul_aff(ec::u256_load(kq + (size_t)pi * 8), Rp));
}
// R_dash = k_i R in a launch of its own: a small batch runs it BESIDE u1 = alpha R of the PDL proofs (both need R only)
__global__ void __launch_bounds__(64) MPE_EC_OCC r4_rbar_kernel(Dim d, const uint32_t* __restrict__ kq, const uint32_t* __restrict__ R, uint32_t* __restrict__ Rbar) {
  MPE_FOREGROUND();
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  ec::aff_store(Rbar + (size_t)pi * 16, mul_aff(ec::u256_load(kq + (size_t)pi * 8), ec::aff_load(R + (size_t)pi * 16)));
}

// ---- Round 5 -------------------------------------------------------------------------------------------------------
// M4 record: S sub-records of 450: proof for peer slot jj (z 0 | u1 64 | u2 80 | u3 208 | s1 272 | s2 297 | s3 361); the last: R_dash
__global__ void idx5_kernel(Dim d, Slab in4, int32_t* __restrict__ sub4_pv, int32_t* __restrict__ rdash_pv) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int P1 = d.S - 1;
  if (g >= d.B * d.PV * d.S * P1) return;
  const int jj = g % P1, r1 = g / P1, i = r1 % d.S, b = r1 / d.S / d.PV;
  const long long r = rec_index(in4, i, b) * d.S;
  sub4_pv[g] = (int32_t)(r + jj); rdash_pv[g] = (int32_t)(r + P1);
}
// my PDL verifications (rounds.rs:546-558), the R_dash sum, S_i and HomoELGamalProof::prove (party_i.rs:768-799)
struct Heg { uint32_t *S, *T, *A3, *z1, *z2; };      // [pi]
__global__ void __launch_bounds__(64) MPE_EC_OCC r5_status_kernel(Dim d, Slab in4, const uint8_t* __restrict__ pdl_ok, int32_t* __restrict__ status,
                                                       uint32_t* __restrict__ bad) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int P1 = d.S - 1, li = pi % d.L, b = pi / d.L;
  const int vo = d.PV == 1 ? 0 : li;
  uint32_t mask = 0;
  for (int i = 0; i < d.S && !mask; ++i)                                                  // `?` stops at the first failing prover
    for (int jj = 0; jj < P1; ++jj)
      if (!pdl_ok[(((size_t)b * d.PV + vo) * d.S + i) * P1 + jj]) mask = 1u << i;
  if (mask) fail(status, bad, pi, 501, mask);
  ec::Jac acc = ec::jac_inf();
  for (int j = 0; j < d.S; ++j) acc = ec::jac_add(acc, ec::jac_from_aff(ec::aff_load(rec_of(in4, j, b) + (size_t)P1 * SUB4)));
  if (!ec::jac_eq_aff(acc, ec::aff_gen())) fail(status, bad, pi, 502, 0);                  // phase5_check_R_dash_sum
}
// S_i and HomoELGamalProof::prove: independent of the verifications above (small batches run it beside them)
__global__ void __launch_bounds__(64) MPE_EC_OCC r5_prove_kernel(Dim d, const uint32_t* __restrict__ R, const uint32_t* __restrict__ sigma_i,
                          const uint32_t* __restrict__ lq, const uint32_t* __restrict__ pedT, const uint32_t* __restrict__ s1_in,
                          const uint32_t* __restrict__ s2_in, Heg h) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2();
  const ec::Aff Rp = ec::aff_load(R + (size_t)pi * 16), T = ec::aff_load(pedT + (size_t)pi * 16);
  const ec::U256 si = ec::u256_load(sigma_i + (size_t)pi * 8), l = ec::u256_load(lq + (size_t)pi * 8);
  const ec::Aff Sp = mul_aff(si, Rp);
  const ec::U256 s1 = ec::sc_reduce(s1_in + (size_t)pi * 8, 8), s2 = ec::sc_reduce(s2_in + (size_t)pi * 8, 8);
  const ec::Aff A3 = mul_aff(s2, Rp), TT = ec::jac_to_aff(ec::jac_add(ec::jac_mul_h2(s1), ec::jac_mul_gen(s2)));
  const ec::Aff hp[7] = {TT, A3, Rp, H, G, T, Sp};
  const ec::U256 e = hash_points(hp, d.enc, d.enc.ord_heg);
  ec::aff_store(h.S + (size_t)pi * 16, Sp);
  ec::aff_store(h.T + (size_t)pi * 16, TT);
  ec::aff_store(h.A3 + (size_t)pi * 16, A3);
  ec::u256_store(h.z1 + (size_t)pi * 8, ec::u256_is_zero(l) ? s1 : ec::sc_add(s1, ec::sc_mul(l, e)));
  ec::u256_store(h.z2 + (size_t)pi * 8, ec::sc_add(s2, ec::sc_mul(si, e)));
}

// ---- Round 6: every HomoELGamalProof; sum S_i == y (party_i.rs:801-848) --------------------------------------------
// M5 record: S_i 0 | T 16 | A3 32 | z1 48 | z2 56
__global__ void __launch_bounds__(64) MPE_EC_OCC r6_kernel(Dim d, Slab in5, const uint32_t* __restrict__ R, const uint32_t* __restrict__ tvec,
                          const uint32_t* __restrict__ y, int32_t* __restrict__ status, uint32_t* __restrict__ bad) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int b = pi / d.L;
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2(), Rp = ec::aff_load(R + (size_t)pi * 16);
  uint32_t mask = 0;
  ec::Jac acc = ec::jac_inf();
  for (int j = 0; j < d.S; ++j) {
    const uint32_t* m = rec_of(in5, j, b);
    const ec::Aff E = ec::aff_load(m), TT = ec::aff_load(m + 16), A3 = ec::aff_load(m + 32),
                  D = ec::aff_load(tvec + ((size_t)j * d.B + b) * 16);
    const ec::Aff hp[7] = {TT, A3, Rp, H, G, D, E};
    const ec::U256 e = hash_points(hp, d.enc, d.enc.ord_heg), z1 = ec::sc_reduce(m + 48, 8), z2 = ec::sc_reduce(m + 56, 8);
    const ec::Jac l1 = ec::jac_add(ec::jac_mul_h2(z1), ec::jac_mul_gen(z2));
    const ec::Jac r1 = ec::jac_add_aff(ec::jac_mul(e, D), TT);
    const ec::Jac l2 = ec::jac_mul(z2, Rp);
    const ec::Jac r2 = ec::jac_add_aff(ec::jac_mul(e, E), A3);
    if (!(ec::jac_eq(l1, r1) && ec::jac_eq(l2, r2))) mask |= 1u << j;
    acc = ec::jac_add_aff(acc, E);
  }
  if (mask) fail(status, bad, pi, 601, mask);
  if (!ec::jac_eq_aff(acc, ec::aff_load(y + (size_t)ks_of(d, b) * 16))) fail(status, bad, pi, 602, 0);
}

// ---- small batches: the same checks with a group of adjacent lanes per item ---------------------------
// One party per lane runs an item's scalar multiplications back to back; with few sessions that leaves most SIMDs
// idle for the whole latency of the chain.  The *_group kernels give every item G lanes: the independent
// multiplications run one per lane (all lanes of a phase execute the same routine on different data), the Jacobian
// results meet in LDS and lane 0 of the group does the additions, comparisons and stores.  With many sessions the
// idle lanes of the groups cost more than the latency they hide (measured at 65 536 sessions: r2a 42 -> 57 ms,
// r3 22 -> 52 ms), so the rounds pick them only when the grouped launch still fits the chip.
struct JacSlots { ec::Jac v[64]; };

__global__ void __launch_bounds__(64) MPE_EC_OCC r2a_group_kernel(Dim d, const int32_t* __restrict__ sub1_rv, const uint32_t* __restrict__ in1,
                           const uint32_t* __restrict__ alpha_full, const uint32_t* __restrict__ kq, const uint32_t* __restrict__ gw,
                           uint32_t* __restrict__ alpha, uint32_t* __restrict__ bpk_in, uint8_t* __restrict__ code) {
  MPE_FOREGROUND();
  // 4 lanes per incoming MessageB: lanes 0..2 do  k_i B | c1 B | c2 B'  (variable base), then  alpha G | z G | z' G
  __shared__ JacSlots vs, fs;
  const int gid = blockIdx.x * 64 + threadIdx.x, rv = gid >> 2, sub = gid & 3, base = (int)threadIdx.x - sub;
  const int P1 = d.S - 1;
  const bool live = rv < d.B * d.L * P1 * 2;
  const int rvc = live ? rv : 0;
  const int v = rvc & 1, pp = rvc >> 1, jj = pp % P1, pi = pp / P1, i = d.loc[pi % d.L], b = pi / d.L, ind = ind_of(i, jj);
  const uint32_t* m = in1 + (size_t)sub1_rv[rvc] * SUB1;
  const ec::U256 al = ec::sc_reduce(alpha_full + (size_t)rvc * 64, 64);
  const ec::Aff Bpk = ec::aff_load(m + 128), BTpk = ec::aff_load(m + 168);
  const ec::Aff R1 = ec::aff_load(m + 144), R2 = ec::aff_load(m + 184);
  ec::Jac res = ec::jac_inf(), fr = ec::jac_inf();
  if (live && sub < 3) {
    const ec::U256 sc = sub == 0 ? ec::u256_load(kq + (size_t)pi * 8) : (sub == 1 ? dlog_challenge(R1, Bpk, d.enc) : dlog_challenge(R2, BTpk, d.enc));
    res = ec::jac_mul(sc, sub == 2 ? BTpk : Bpk);
    const ec::U256 fk = sub == 0 ? al : ec::sc_reduce(m + (sub == 1 ? 160 : 200), 8);
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
  r2a_finish(d, rv, v, pp, b, ind, Bpk, good, gw, bpk_in, code);
}

// G lanes per party (a power of two >= 2 S): lane 2j | 2j+1 does z1_j G | z2_j H, lane j also e_j com_j
__global__ void __launch_bounds__(64) MPE_EC_OCC r3_group_kernel(Dim d, int G, Slab in2, uint32_t* __restrict__ dinv, int32_t* __restrict__ status,
                                                      uint32_t* __restrict__ bad) {
  MPE_FOREGROUND();
  __shared__ JacSlots vs, fs;
  const int gid = blockIdx.x * 64 + threadIdx.x, pi = gid / G, sub = gid % G, base = (int)threadIdx.x - sub;
  const bool live = pi < d.B * d.L;
  const int b = live ? pi / d.L : 0;
  const ec::Aff Gp = ec::aff_gen(), H = ec::aff_h2();
  ec::Jac res = ec::jac_inf(), fr = ec::jac_inf();
  if (live && sub < d.S) {
    const uint32_t* m = rec_of(in2, sub, b);
    const ec::Aff C = ec::aff_load(m + 64), a1 = ec::aff_load(m + 32), a2 = ec::aff_load(m + 48);
    const ec::Aff hp[5] = {Gp, H, C, a1, a2};
    res = ec::jac_mul(hash_points(hp, d.enc, d.enc.ord_pedersen), C);
  }
  if (live && sub < 2 * d.S) {
    const uint32_t* m = rec_of(in2, sub >> 1, b);
    fr = ec::jac_mul_fixed(ec::sc_reduce(m + ((sub & 1) ? 88 : 80), 8), sub & 1);
  }
  vs.v[threadIdx.x] = res;
  fs.v[threadIdx.x] = fr;
  __syncthreads();
  if (!live || sub) return;
  ec::U256 sum = ec::u256_zero();
  bool com_ok = true, ped_ok = true;
  for (int j = 0; j < d.S; ++j) {
    const uint32_t* m = rec_of(in2, j, b);
    sum = ec::sc_add(sum, ec::sc_reduce(m, 8));
    com_ok = com_ok && words_eq(m + 8, m + 64, 16);
    const ec::Aff a1 = ec::aff_load(m + 32), a2 = ec::aff_load(m + 48);
    const ec::Jac lhs = ec::jac_add(fs.v[base + 2 * j], fs.v[base + 2 * j + 1]);
    const ec::Jac rhs = ec::jac_add_aff(ec::jac_add_aff(vs.v[base + j], a1), a2);
    ped_ok = ped_ok && ec::jac_eq(lhs, rhs);
  }
  r3_finish(pi, com_ok, ped_ok, sum, dinv, status, bad);
}

// G lanes per party (a power of two >= 3 S): lane 3j+k does e D_j | z2_j R | e E_j; lane 2j+k does z1_j H | z2_j G
__global__ void __launch_bounds__(64) MPE_EC_OCC r6_group_kernel(Dim d, int G, Slab in5, const uint32_t* __restrict__ R, const uint32_t* __restrict__ tvec,
                          const uint32_t* __restrict__ y, int32_t* __restrict__ status, uint32_t* __restrict__ bad) {
  __shared__ JacSlots vs, fs;
  const int gid = blockIdx.x * 64 + threadIdx.x, pi = gid / G, sub = gid % G, base = (int)threadIdx.x - sub;
  const bool live = pi < d.B * d.L;
  const int pic = live ? pi : 0, b = pic / d.L;
  const ec::Aff Gp = ec::aff_gen(), H = ec::aff_h2(), Rp = ec::aff_load(R + (size_t)pic * 16);
  ec::Jac res = ec::jac_inf(), fr = ec::jac_inf();
  if (live && sub < 3 * d.S) {
    const int j = sub / 3, kind = sub % 3;
    const uint32_t* m = rec_of(in5, j, b);
    const ec::Aff E = ec::aff_load(m), TT = ec::aff_load(m + 16), A3 = ec::aff_load(m + 32),
                  D = ec::aff_load(tvec + ((size_t)j * d.B + b) * 16);
    const ec::Aff hp[7] = {TT, A3, Rp, H, Gp, D, E};
    const ec::U256 e = hash_points(hp, d.enc, d.enc.ord_heg);
    res = ec::jac_mul(kind == 1 ? ec::sc_reduce(m + 56, 8) : e, kind == 0 ? D : (kind == 1 ? Rp : E));
  }
  if (live && sub < 2 * d.S) {
    const uint32_t* m = rec_of(in5, sub >> 1, b);
    fr = ec::jac_mul_fixed(ec::sc_reduce(m + ((sub & 1) ? 56 : 48), 8), (sub & 1) ? 0 : 1);
  }
  vs.v[threadIdx.x] = res;
  fs.v[threadIdx.x] = fr;
  __syncthreads();
  if (!live || sub) return;
  uint32_t mask = 0;
  ec::Jac acc = ec::jac_inf();
  for (int j = 0; j < d.S; ++j) {
    const uint32_t* m = rec_of(in5, j, b);
    const ec::Aff E = ec::aff_load(m), TT = ec::aff_load(m + 16), A3 = ec::aff_load(m + 32);
    const ec::Jac l1 = ec::jac_add(fs.v[base + 2 * j], fs.v[base + 2 * j + 1]);
    const ec::Jac r1 = ec::jac_add_aff(vs.v[base + 3 * j], TT);
    const ec::Jac r2 = ec::jac_add_aff(vs.v[base + 3 * j + 2], A3);
    if (!(ec::jac_eq(l1, r1) && ec::jac_eq(vs.v[base + 3 * j + 1], r2))) mask |= 1u << j;
    acc = ec::jac_add_aff(acc, E);
  }
  if (mask) fail(status, bad, pi, 601, mask);
  if (!ec::jac_eq_aff(acc, ec::aff_load(y + (size_t)ks_of(d, b) * 16))) fail(status, bad, pi, 602, 0);
}

// ---- Round 7: phase7_local_sig -> PartialSignature (party_i.rs:850-871) -------------------------------------------------
__global__ void __launch_bounds__(64) MPE_EC_OCC r7_kernel(Dim d, const uint32_t* __restrict__ msg, const uint32_t* __restrict__ R,
                          const uint32_t* __restrict__ kq, const uint32_t* __restrict__ sigma_i, uint32_t* __restrict__ mq,
                          uint32_t* __restrict__ rq, uint32_t* __restrict__ s_i, int fault_step, uint32_t fault_mask) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int b = pi / d.L;
  const ec::U256 m = ec::sc_reduce(msg + (size_t)b * 8, 8), r = ec::sc_reduce(R + (size_t)pi * 16, 8);
  ec::u256_store(mq + (size_t)pi * 8, m);
  ec::u256_store(rq + (size_t)pi * 8, r);
  ec::U256 si = ec::sc_add(ec::sc_mul(m, ec::u256_load(kq + (size_t)pi * 8)), ec::sc_mul(r, ec::u256_load(sigma_i + (size_t)pi * 8)));
  if (fault_step == 7 && ((fault_mask >> d.loc[pi % d.L]) & 1u)) si = ec::sc_add(si, si);              // test.rs:679-686
  ec::u256_store(s_i + (size_t)pi * 8, si);
}
// SignManual::complete -> output_signature + verify (party_i.rs:873-936); outputs are [L][B], zero unless status == 0
__global__ void __launch_bounds__(64) MPE_EC_OCC complete_kernel(Dim d, Slab in6, const uint32_t* __restrict__ R, const uint32_t* __restrict__ mq,
                          const uint32_t* __restrict__ rq, const uint32_t* __restrict__ s_i, const uint32_t* __restrict__ y,
                          int32_t* __restrict__ status, uint32_t* __restrict__ bad, uint32_t* __restrict__ r_out,
                          uint32_t* __restrict__ s_out, int32_t* __restrict__ recid_out) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const int li = pi % d.L, i = d.loc[li], b = pi / d.L;
  ec::U256 s = ec::u256_load(s_i + (size_t)pi * 8);
  for (int j = 0; j < d.S; ++j) if (j != i) s = ec::sc_add(s, ec::sc_reduce(rec_of(in6, j, b), 8));
  const ec::U256 m = ec::u256_load(mq + (size_t)pi * 8), r = ec::u256_load(rq + (size_t)pi * 8);
  const ec::U256 ry = ec::sc_reduce(R + (size_t)pi * 16 + 8, 8);
  int recid = (int)(ry.w[0] & 1u);
  const ec::U256 neg = ec::sc_neg(s);
  {  // if s > q - s: s = q - s, recid ^= 1
    bool gt = false;
    for (int j = 7; j >= 0; --j) { if (s.w[j] != neg.w[j]) { gt = s.w[j] > neg.w[j]; break; } }
    if (gt) { s = neg; recid ^= 1; }
  }
  bool okv = !ec::u256_is_zero(s);
  if (okv) {
    const ec::U256 bi = ec::sc_inv(s), u1 = ec::sc_mul(m, bi), u2 = ec::sc_mul(r, bi);
    const ec::Aff V = ec::jac_to_aff(ec::jac_add(ec::jac_mul_gen(u1), ec::jac_mul(u2, ec::aff_load(y + (size_t)ks_of(d, b) * 16))));
    okv = !V.inf && ec::u256_eq(ec::sc_reduce(V.x.w, 8), r);
  }
  const size_t nPI = (size_t)d.B * d.L;
  if (!okv) fail(status + 8 * nPI, bad + 8 * nPI, pi, 701, 0);
  const bool good = resolve_status(status, nPI, pi, NR - 1, nullptr, nullptr) == 0;
  const size_t o = (size_t)li * d.B + b;
  ec::u256_store(r_out + o * 8, good ? r : ec::u256_zero());
  ec::u256_store(s_out + o * 8, good ? s : ec::u256_zero());
  recid_out[o] = good ? recid : 0;
}
// [L][B] views of the per-party state
__global__ void result_kernel(Dim d, const int32_t* __restrict__ status, const uint32_t* __restrict__ bad, const uint32_t* __restrict__ R,
                              int32_t* __restrict__ st_out, uint32_t* __restrict__ bad_out, uint32_t* __restrict__ R_out) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= d.B * d.L) return;
  const size_t o = (size_t)(pi % d.L) * d.B + pi / d.L;
  uint32_t bm = 0;
  const int stv = resolve_status(status, (size_t)d.B * d.L, pi, NR - 1, bad, &bm);
  if (st_out) st_out[o] = stv;
  if (bad_out) bad_out[o] = bm;
  if (R_out) for (int j = 0; j < 16; ++j) R_out[o * 16 + j] = R[(size_t)pi * 16 + j];
}
// the lock-step composition: a session's status = the smallest non-zero party status; the signature is party 0's
__global__ void sign_finish_kernel(int B, int L, int b0, const int32_t* __restrict__ pst, const uint32_t* __restrict__ pr,
                                   const uint32_t* __restrict__ ps, const int32_t* __restrict__ prec, const uint32_t* __restrict__ pR,
                                   uint32_t* __restrict__ r_out, uint32_t* __restrict__ s_out, int32_t* __restrict__ recid_out,
                                   uint32_t* __restrict__ R_out, int32_t* __restrict__ status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int st = 0;
  for (int li = 0; li < L; ++li) { const int s = pst[(size_t)li * B + b]; if (s && (!st || s < st)) st = s; }
  const size_t o = (size_t)b0 + b;
  for (int j = 0; j < 8; ++j) { r_out[o * 8 + j] = st ? 0u : pr[(size_t)b * 8 + j]; s_out[o * 8 + j] = st ? 0u : ps[(size_t)b * 8 + j]; }
  recid_out[o] = st ? 0 : prec[b];
  if (R_out) for (int j = 0; j < 16; ++j) R_out[o * 16 + j] = pR[(size_t)b * 16 + j];
  status[o] = st;
}

}  // namespace gg
}  // namespace mpe

// ======================================================================================================================
// the session object: one batch of B sessions x L local parties, carried from round to round
// ======================================================================================================================
struct mpe_gg20_session {
  mpe_ctx* ctx = nullptr;
  const mpe_gg20_keys* K = nullptr;
  int B = 0, L = 0, dedup = 0, next_round = 0;
  bool failed = false;             // a round call returned an error (MPE_E_NOMEM, a HIP failure): the batch cannot go on; rearm / abort start afresh
  mpe::gg::Dim d{};
  mpe_gg20_nonces Z{};
  void* mem = nullptr;
  size_t mem_bytes = 0;
  bool from_cache = false;
  mpe::gg::Idx ix{};
  // state that lives across rounds (what the reference's RoundN structs carry)
  uint32_t *kq = nullptr, *gq = nullptr, *w = nullptr, *k64 = nullptr, *g_gamma = nullptr, *com = nullptr, *c_a = nullptr, *beta = nullptr;
  uint32_t *ca_all = nullptr, *com_all = nullptr, *bpk_in = nullptr, *delta_i = nullptr, *sigma_i = nullptr, *lq = nullptr, *pedT = nullptr;
  uint32_t *tvec = nullptr, *dinv = nullptr, *R = nullptr, *Rbar = nullptr, *mq = nullptr, *rq = nullptr, *s_i = nullptr, *bad = nullptr;
  uint32_t *sig_r = nullptr, *sig_s = nullptr;
  uint32_t* miu = nullptr;         // [pp][64] the plaintexts of the incoming w_i MessageBs before reduction (LocalStatePhase6::miu, blame.rs:227-234)
  int32_t *status = nullptr, *sig_recid = nullptr;
  int32_t *sub0_vi = nullptr, *sub4_pv = nullptr, *rdash_pv = nullptr;
  uint8_t *ok_vi = nullptr, *ok_pv = nullptr;
  // lock-step signing of a small batch (mpe_gg20_sign: every signer local, the rounds chained inside the library): round 0 already inverts
  // the ciphertexts round 1's verifiers will need, beside the tail of the range proofs (round1_inversion_ahead)
  bool lockstep = false, cinv_ahead = false;
  uint32_t* cinv_pre = nullptr;    // [vi][128] c^-1 mod N^2 of every (verifier, sender, statement) of round 1; small batches only
  uint8_t* cinv_ok_pre = nullptr;  // [vi]
  // ... and round 2 starts the beta^N mod N^2 of round 4's PDL proofs (the prover's own nonce under its own key: no input of any round)
  // as a BACKGROUND launch — wave priority 0 beside the decryption ladder at 2 and the EC kernels of rounds 2 and 3 at 1 (mpe_sched.h)
  uint32_t* ped_pre = nullptr;     // [pi][48] l H | s1 G | s2 H of Round 2 (ped_ahead_kernel, Round 0)
  bool pdl_ahead = false, pdl_phase1 = false;      // pdl_phase1: the first of its two ladders already ran between rounds 0 and 1
  uint32_t* pdl_bn = nullptr;      // [pp][128]
  uint32_t* pdl_scratch = nullptr; // modexp_nn_scratch_words(pp)
  int fault_step = 0;              // fault injection of the reference's tests (gg_2020/test.rs:282-289,458-465,679-686): 5 / 6 / 7
  uint32_t fault_mask = 0;         // signer ordinals that double their delta_i / sigma_i / s_i
  char* tmp = nullptr;             // per-round scratch (dense outputs of the composites before they are packed)
  size_t tmp_bytes = 0;
};

namespace mpe {
namespace gg {

// option gg20_trace (mpe_ctx_set_option): synchronise after every step and report it on stderr (debug aid)
static void gg_trace(const mpe_ctx* ctx, hipStream_t st, const char* what, int rc) {
  if (!ctx->gg20_trace) return;
  const hipError_t e = hipStreamSynchronize(st);
  fprintf(stderr, "[gg20] %-28s rc=%d sync=%s\n", what, rc, hipGetErrorString(e));
  fflush(stderr);
}
#define GG_LAUNCH(kernel, nitems, ...)                                                                   \
  do {                                                                                                    \
    if (rc == MPE_OK && (nitems) > 0)                                                                     \
      hipLaunchKernelGGL(kernel, dim3(blocks_for((int)(nitems), 64)), dim3(64), 0, st, __VA_ARGS__);      \
    gg_trace(s->ctx, st, #kernel, rc);                                                                            \
  } while (0)

struct Bump {
  char* base; size_t off = 0;
  explicit Bump(char* b) : base(b) {}
  void* take(size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return base ? base + o : nullptr; }
  uint32_t* w(size_t words) { return (uint32_t*)take(words * 4); }
  int32_t* i(size_t count) { return (int32_t*)take(count * 4); }
  uint8_t* f(size_t count) { return (uint8_t*)take(count); }
};
struct Counts { size_t nPI, nAP, nVI, nMB, nPP, nPV, SB; };
static Counts counts_of(const Dim& d) {
  const size_t P1 = d.S - 1;
  Counts c;
  c.nPI = (size_t)d.B * d.L; c.nAP = c.nPI * d.n; c.nPP = c.nPI * P1; c.nMB = c.nPP * 2; c.nVI = c.nPP * d.V * d.n;
  c.nPV = (size_t)d.B * d.PV * d.S * P1; c.SB = (size_t)d.S * d.B;
  return c;
}
static size_t tmp_bytes_of(const Counts& c) {
  const size_t r0 = c.nAP * 250 * 4 + 4096;
  const size_t r1 = c.nVI * 5 + c.nMB * (8 + 8 + 128 + 16 + 16 + 8 + 16 + 16 + 8) * 4 + 8192;
  const size_t r2 = c.nMB * (1 + 64 + 8) * 4 + c.nMB + c.nPI * 72 * 4 + 8192;
  const size_t r4 = c.nPP * 450 * 4 + 4096;
  const size_t r5 = c.nPV * 9 + c.nPI * 64 * 4 + 8192;
  size_t m = r0;
  if (r1 > m) m = r1;
  if (r2 > m) m = r2;
  if (r4 > m) m = r4;
  if (r5 > m) m = r5;
  return m + 64 * 256;                 // alignment slack of the bump allocator
}
// assigns every state array (base == nullptr: only sizes)
static size_t layout(mpe_gg20_session* s, char* base) {
  const Dim& d = s->d;
  const Counts c = counts_of(d);
  Bump m(base);
  Idx& x = s->ix;
  x.kown_pi = m.i(c.nPI);
  x.pi_ap = m.i(c.nAP); x.kown_ap = m.i(c.nAP); x.st_ap = m.i(c.nAP);
  x.ca_vi = m.i(c.nVI); x.kpub_vi = m.i(c.nVI); x.st_vi = m.i(c.nVI);
  x.ca_mb = m.i(c.nMB); x.kpub_mb = m.i(c.nMB); x.kown_mb = m.i(c.nMB);
  x.pi_pp = m.i(c.nPP); x.kown_pp = m.i(c.nPP); x.st_pp = m.i(c.nPP);
  x.ca_pv = m.i(c.nPV); x.pi_pv = m.i(c.nPV); x.kpub_pv = m.i(c.nPV); x.st_pv = m.i(c.nPV);
  s->kq = m.w(c.nPI * 8); s->gq = m.w(c.nPI * 8); s->w = m.w(c.nPI * 8); s->k64 = m.w(c.nPI * 64);
  s->g_gamma = m.w(c.nPI * 16); s->com = m.w(c.nPI * 8); s->c_a = m.w(c.nPI * 128); s->beta = m.w(c.nMB * 8);
  s->ca_all = m.w(c.SB * 128); s->com_all = m.w(c.SB * 8); s->bpk_in = m.w(c.nPP * 16);
  s->delta_i = m.w(c.nPI * 8); s->sigma_i = m.w(c.nPI * 8); s->lq = m.w(c.nPI * 8); s->pedT = m.w(c.nPI * 16);
  s->tvec = m.w(c.SB * 16); s->dinv = m.w(c.nPI * 8); s->R = m.w(c.nPI * 16); s->Rbar = m.w(c.nPI * 16);
  s->mq = m.w(c.nPI * 8); s->rq = m.w(c.nPI * 8); s->s_i = m.w(c.nPI * 8); s->bad = m.w(c.nPI * NR);
  s->sig_r = m.w(c.nPI * 8); s->sig_s = m.w(c.nPI * 8); s->miu = m.w(c.nPP * 64);
  s->status = m.i(c.nPI * NR); s->sig_recid = m.i(c.nPI);
  // index tables and verdicts of the two verification rounds
  s->sub0_vi = m.i(c.nVI); s->ok_vi = m.f(c.nVI); s->sub4_pv = m.i(c.nPV); s->rdash_pv = m.i(c.nPV); s->ok_pv = m.f(c.nPV);
  const bool small = s->ctx->allow_par && c.nVI <= (size_t)s->ctx->par_items;       // the batches whose composites fork
  s->cinv_pre = small ? m.w(c.nVI * 128) : nullptr; s->cinv_ok_pre = small ? m.f(c.nVI) : nullptr;
  s->ped_pre = m.w(c.nPI * 48);
  s->pdl_bn = small ? m.w(c.nPP * 128) : nullptr; s->pdl_scratch = small ? m.w(modexp_nn_scratch_words(c.nPP)) : nullptr;
  s->tmp_bytes = tmp_bytes_of(c);
  s->tmp = (char*)m.take(s->tmp_bytes);
  return m.off;
}
static Slab slab_of(const mpe_gg20_session* s, const uint32_t* p, const int64_t* h_off, int round) {
  Slab sl;
  sl.p = p; sl.W = msg_words(s->d.S, s->d.n, round);
  for (int j = 0; j < 8; ++j) sl.off[j] = j < s->d.S ? (h_off ? (long long)h_off[j] : (long long)j * s->B) : 0;
  return sl;
}
static int round_enter(mpe_gg20_session* s, int round, const void* in, const void* out, bool need_in, bool need_out) {
  if (!s || (need_in && !in) || (need_out && !out)) return MPE_E_ARG;
  // a round call that FAILED (MPE_E_NOMEM, a HIP failure) may have written half of this round's state: the same round must not be
  // entered again on it — mpe_gg20_session_rearm / _abort are the only ways on (include/mpecdsa_hip.h: "the batch cannot go on")
  if (s->failed) { mpe_set_error_msg("gg20: a round call of this batch failed (or the batch was aborted): re-arm the session"); return MPE_E_ARG; }
  if (s->next_round != round) { mpe_set_error_msg("gg20: rounds must be run in order"); return MPE_E_ARG; }
  s->d.enc = s->ctx->enc;          // the transcript conventions are the context's, read when a round starts
  return MPE_OK;
}
static int round_exit(mpe_gg20_session* s, int rc, const char* what) {
  if (rc != MPE_OK) { s->failed = true; return rc; }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { s->failed = true; mpe_set_error(what, e); return MPE_E_HIP; }
  s->next_round++;
  return MPE_OK;
}
#define STAT(r) (s->status + (size_t)(r) * c.nPI)
#define BADR(r) (s->bad + (size_t)(r) * c.nPI)
#define PACK(nitems, per, nsub, sub0, subw, dst_off, src, words)                                                              \
  GG_LAUNCH(pack_field_kernel, (size_t)(nitems) * (words), (int)(nitems), (int)(per), d.L, d.B, (int)(nsub), (int)(sub0), (int)(subw),  \
            (int)(dst_off), (const uint32_t*)(src), (int)(words), s->status, s->next_round, d_out)

// does round 1 run the ladders of its verifications and of its MessageBs in ONE launch (round1_merged_ladders)?  Round 0 asks too:
// only then is the inversion of the ciphertexts in front of a ladder that everything waits for
static bool round1_merges(const mpe_ctx* ctx, const Counts& c) {
  const bool par = ctx->allow_par && (int)c.nVI <= ctx->par_items;
  const size_t resident_groups = (size_t)ctx->cus * ctx->modexp_waves_per_cu * 16 / ctx->device_share;
  return ctx->merge_r1 && ctx->use_pair && ctx->use_multiexp && c.nVI > 0 && c.nMB > 0 && c.nVI + c.nMB < ((size_t)1 << 30) &&
         (!par || 4 * (c.nVI + c.nMB) > (size_t)ctx->merge_r1_quarters * resident_groups);
}
// every signer local, local slot l = signer ordinal l (mpe_gg20_sign): ca_all[ind][b] = c_a[b][ind] — what round 1's gather_field_kernel
// will read out of round 0's message slab, before the slab is written
__global__ void ca_all_from_local_kernel(int S, int B, const uint32_t* __restrict__ c_a, uint32_t* __restrict__ ca_all) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)S * B * 128) return;
  const int w = (int)(g % 128);
  const size_t jb = g / 128;
  const int ind = (int)(jb / B), b = (int)(jb % B);
  ca_all[g] = c_a[((size_t)b * S + ind) * 128 + w];
}
static size_t ws_need_inversion_ahead(const mpe_paillier* pk, size_t nVI) { return nVI * (128 + 128 + 1) * 4 + modinv_ws_words(pk->ms_nn, (int)nVI) * 4 + 16384; }

// ---- Round0::proceed (rounds.rs:68-104) ------------------------------------------------------------------------------
static int round0(mpe_gg20_session* s, uint32_t* d_out, hipStream_t st) {
  int rc = round_enter(s, 0, nullptr, d_out, false, true);
  if (rc != MPE_OK) return rc;
  mpe_ctx* ctx = s->ctx; const mpe_gg20_keys* K = s->K; const Dim& d = s->d; const Counts c = counts_of(d); const mpe_gg20_nonces& Z = s->Z;
  const int n = d.n;
  (void)hipMemsetAsync(d_out, 0, c.nPI * (size_t)msg_words(d.S, n, 0) * 4, st);
  // (r0_kernel — k_i, g^gamma_i, the commitments: 0.9 ms of EC — is launched BEHIND the fork below: the x^N ladders of a small batch need
  //  the nonces only and start at once; the encryption's tail waits for k_i through an event)
  // small batches: the encryption of k_i runs beside the first half of the range proofs (the proofs need c only for their
  // transcript hash); both composites draw from ONE workspace reservation
  const bool par = ctx->allow_par && (int)c.nAP <= ctx->par_items;
  const size_t nXN = c.nPI + c.nAP;
  // (round 6) lock-step signing: the inversion of the ciphertexts that round 1's verifiers need (6.7 ms of dependent short launches at
  // 1 024 sessions, in front of the ladder the whole round waits for) starts HERE, on the forked stream behind the encryption, beside the
  // tail of the range proofs (transcript hash, r^e, the linear responses: 4 ms on the caller's stream).  Same inputs, same kernels, same
  // verdicts — round 1 finds them done.  Only where the message slab cannot change between the two rounds: mpe_gg20_sign.
  bool all_local_in_order = d.L == d.S;           // local slot l IS signer ordinal l (ca_all_from_local_kernel)
  for (int l = 0; l < d.L && all_local_in_order; ++l) all_local_in_order = d.loc[l] == l;
  const bool ahead = par && s->lockstep && s->cinv_pre && all_local_in_order && !ctx->no_r1_inversion_ahead && round1_merges(ctx, c);
  s->cinv_ahead = false;
  if (par && rc == MPE_OK) {
    rc = ws_reserve(ctx, ws_need_encrypt((int)c.nPI) + ws_need_alice_generate((int)c.nAP) + nXN * (64 + 128 + 1 + CRT_WS_WORDS) * 4 + 65536 +
                         (ahead ? ws_need_inversion_ahead(K->pub, c.nVI) : 0), st);
    if (rc == MPE_OK) ctx->ws_hold++;
  }
  const bool held = par && rc == MPE_OK;
  // ... and every x^N the key holders compute this round (the randomness of MessageA.c, the beta of each range proof) goes
  // through ONE launch: two concurrent launches of a few hundred waves each start on the same SIMDs of every CU and take
  // 15 ms where one launch of 1 024 waves takes 11.5
  // (round 6) ... on the FORKED stream, in front of the encryption's tail that needs it: the range proofs' N~ side (four fixed-base
  // powers per proof) starts at once on the caller's stream beside the ladders instead of behind them; the proofs wait for beta^N only
  // where they multiply it in (ev_mid)
  const uint32_t *rn_pre = nullptr, *bn_pre = nullptr;
  Fork g(ctx, st, 2, held, 2);
  GG_LAUNCH(r0_kernel, c.nPI, d, K->x, Z.k, Z.gamma, Z.blind, s->kq, s->gq, s->w, s->k64, s->g_gamma, s->com, STAT(0), BADR(0));
  if (g.on) (void)hipEventRecord(ctx->ev_fork[0], st);                             // "k_i is there" (waited for in front of the encryption's tail)
  GG_LAUNCH(ped_ahead_kernel, c.nPI, d, Z.l, Z.ped_s1, Z.ped_s2, s->ped_pre);      // Round 2's nonce-only points, beside this round's ladders
  hipEvent_t xn_ready = nullptr;
  if (held && ctx->merge_xn) {
    hipStream_t sx = g.s(1);
    uint32_t* xs = ws_array<uint32_t>(ctx, nXN * 64);
    uint32_t* xn = ws_array<uint32_t>(ctx, nXN * 128);
    int32_t* kx = ws_array<int32_t>(ctx, nXN);
    if (!xs || !xn || !kx) rc = MPE_E_NOMEM;
    if (rc == MPE_OK) {
      (void)hipMemcpyAsync(xs, Z.r_a, c.nPI * 64 * 4, hipMemcpyDeviceToDevice, sx);
      (void)hipMemcpyAsync(xs + c.nPI * 64, Z.al_beta, c.nAP * 64 * 4, hipMemcpyDeviceToDevice, sx);
      (void)hipMemcpyAsync(kx, s->ix.kown_pi, c.nPI * 4, hipMemcpyDeviceToDevice, sx);
      (void)hipMemcpyAsync(kx + c.nPI, s->ix.kown_ap, c.nAP * 4, hipMemcpyDeviceToDevice, sx);
      rc = modexp_nn(ctx, K->prv, (int)nXN, key_selector(K->prv, kx), rows(xs, 64, nullptr, 64), key_rows(K->prv, K->prv->N, 64, kx), 64, true, xn, sx, true);
      rn_pre = xn; bn_pre = xn + c.nPI * 128;
      if (rc == MPE_OK && g.on) { (void)hipEventRecord(ctx->ev_mid, sx); xn_ready = ctx->ev_mid; }
    }
  }
  if (g.on) (void)hipStreamWaitEvent(g.s(1), ctx->ev_fork[0], 0);                  // k_i (r0_kernel on the caller's stream): long done by now
  if (rc == MPE_OK) rc = paillier_encrypt(ctx, K->prv, (int)c.nPI, s->ix.kown_pi, s->k64, Z.r_a, s->c_a, true, g.s(1), rn_pre);      // MessageA.c
  gg_trace(s->ctx, st, "encrypt k", rc);
  hipEvent_t c_ready = nullptr;
  if (ahead && held && g.on && xn_ready && rc == MPE_OK) {
    hipStream_t sx = g.s(1);
    // the proofs wait for the CIPHERTEXTS from here on (their transcript hash reads them; beta^N came earlier on the same stream), not for
    // whatever else this stream gets: the event moves behind the encryption's two short tail kernels
    (void)hipEventRecord(ctx->ev_mid, sx); c_ready = ctx->ev_mid;
    hipLaunchKernelGGL(ca_all_from_local_kernel, dim3((unsigned)((c.SB * 128 + 255) / 256)), dim3(256), 0, sx, d.S, d.B, s->c_a, s->ca_all);
    Seq q{ctx, sx, (int)c.nVI};
    const Rows ksel = sel_of(s->ix.kpub_vi, K->pub->nkeys);
    uint8_t* ok = q.flags();
    uint32_t* cred = q.modmul(K->pub->ms_nn, ksel, rows(s->ca_all, 128, s->ix.ca_vi), rows(K->pub->ms_nn->one_words, 0, nullptr, 1));
    uint32_t* cinv = q.modinv(K->pub->ms_nn, ksel, rows(cred, 128), ok);
    rc = q.rc;
    if (rc == MPE_OK) {
      (void)hipMemcpyAsync(s->cinv_pre, cinv, c.nVI * 128 * 4, hipMemcpyDeviceToDevice, sx);
      (void)hipMemcpyAsync(s->cinv_ok_pre, ok, c.nVI, hipMemcpyDeviceToDevice, sx);
      s->cinv_ahead = true;
    }
    gg_trace(s->ctx, sx, "round 1's inversion, ahead", rc);
  }
  Bump t(s->tmp);
  mpe_alice_proof ap{t.w(c.nAP * 64), t.w(c.nAP * 8), t.w(c.nAP * 64), t.w(c.nAP * 25), t.w(c.nAP * 89)};
  mpe_alice_nonces an{Z.al_alpha, Z.al_beta, Z.al_gamma, Z.al_rho};
  if (rc == MPE_OK)
    rc = alice_generate(ctx, K->prv, K->stm, (int)c.nAP, s->ix.kown_ap, s->ix.st_ap, rows(s->kq, 8, s->ix.pi_ap),
                        rows(s->c_a, 128, s->ix.pi_ap), rows(Z.r_a, 64, s->ix.pi_ap), &an, &ap, st, &g, bn_pre, c_ready ? c_ready : xn_ready, c_ready);
  g.join();                                       // (again: with c_ready the proofs waited for the event only)
  // ... and the FIRST of the two ladders behind the PDL proofs' beta^N (round 2 starts the rest, round 4 needs it): at priority 0 on the
  // stream of the range proofs' shortest branch, from the moment the ciphertexts are there — the 5 ms before round 1's ladder in which
  // the chip runs an inversion and a hash
  s->pdl_phase1 = false;
  if (rc == MPE_OK && c_ready && s->pdl_bn && ctx->use_prio && !ctx->no_pdl_ahead && ctx->use_pair && ctx->use_crt && ctx->use_pown && c.nPP > 0 &&
      (int)c.nPP <= ctx->par_items) {
    hipStream_t sa = ctx->aux[1];
    (void)hipStreamWaitEvent(sa, c_ready, 0);
    const int keep = ctx->ladder_prio;
    ctx->ladder_prio = 0;
    const int rca = modexp_nn(ctx, K->prv, (int)c.nPP, sel_of(s->ix.kown_pp, K->prv->nkeys), rows(Z.pdl_beta, 64, nullptr, 64),
                              tab_rows(K->prv->N, 64, s->ix.kown_pp, K->prv->nkeys), 64, true, s->pdl_bn, sa, true, s->pdl_scratch, 1);
    ctx->ladder_prio = keep;
    if (rca == MPE_OK) s->pdl_phase1 = true; else rc = rca;
    gg_trace(s->ctx, sa, "PDL beta^N, first ladder, ahead", rc);
  }
  if (held) ctx->ws_hold--;
  gg_trace(s->ctx, st, "alice_generate", rc);
  PACK(c.nAP, n, n + 1, 0, SUB0, 0, ap.z, 64); PACK(c.nAP, n, n + 1, 0, SUB0, 64, ap.e, 8); PACK(c.nAP, n, n + 1, 0, SUB0, 72, ap.s, 64);
  PACK(c.nAP, n, n + 1, 0, SUB0, 136, ap.s1, 25); PACK(c.nAP, n, n + 1, 0, SUB0, 161, ap.s2, 89);
  PACK(c.nPI, 1, n + 1, n, SUB0, 0, s->c_a, 128); PACK(c.nPI, 1, n + 1, n, SUB0, 128, s->com, 8);
  return round_exit(s, rc, "gg20 round0");
}

// dst[i][0..words) = row i of src (zero-extended from src.words): operands of different composites side by side for ONE launch
__global__ void gather_rows_kernel(int n, Rows src, int src_words, int words, uint32_t* __restrict__ dst) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * words) return;
  const int w = (int)(g % words), i = (int)(g / words);
  dst[g] = w < src_words ? row_of(src, i)[w] : 0u;
}
// Round 1, large batches: the two-base ladders of the range-proof verifications, s^N (c^-1)^e (range_proofs.rs:134-141), and of the
// MessageB ciphertexts, r^N c_a^b (mta/mod.rs:133-145), have the same shape — a 2048-bit base to the public exponent N times a
// 4096-bit base to a 256-bit exponent, modulo N^2 of a peer's key — and go through ONE launch of the pair kernel: 12 + 4 items per
// session instead of launches of 12 and of 4, so that e.g. 4 096 sessions are 65 536 items = two full passes of the chip where the
// separate launches took 1.5 + 0.5 passes rounded up to whole trips (the pipelined engine's super-batches live at these sizes).
// Operands are gathered into contiguous rows first (800 bytes per item against a ladder of ~2 500 multiplications modulo N^2).
static size_t ws_need_round1_merged(const mpe_paillier* pk, size_t nVI, size_t nMB) {
  return (nVI + nMB) * (size_t)(64 + 128 + 8 + 1 + 128) * 4 + nVI * (128 + 128 + 1) * 4 + modinv_ws_words(pk->ms_nn, (int)nVI) * 4 + 65536;
}
static int round1_merged_ladders(mpe_ctx* ctx, const mpe_paillier* pk, size_t nVI, const int32_t* kpub_vi, Rows cipher_vi, Rows s_vi, Rows e_vi,
                                 size_t nMB, const int32_t* kpub_mb, Rows ca_mb, const uint32_t* bsel, const uint32_t* mb_r,
                                 const uint32_t** m_vi, const uint8_t** inv_ok_vi, const uint32_t** x_mb, hipStream_t st,
                                 const uint32_t* cinv_pre = nullptr, const uint8_t* cinv_ok_pre = nullptr) {
  const size_t n = nVI + nMB;
  Seq q{ctx, st, (int)nVI};
  const Rows ksel = sel_of(kpub_vi, pk->nkeys);
  const uint8_t* inv_ok = cinv_ok_pre;
  const uint32_t* cinv = cinv_pre;
  if (!cinv_pre) {                                                                                         // (round 0 did it: round1_inversion_ahead)
    uint8_t* ok = q.flags();
    uint32_t* cred = q.modmul(pk->ms_nn, ksel, cipher_vi, rows(pk->ms_nn->one_words, 0, nullptr, 1));      // c mod N^2
    cinv = q.modinv(pk->ms_nn, ksel, rows(cred, 128), ok);                                                 // (c^e)^-1 = (c^-1)^e
    inv_ok = ok;
  }
  if (q.rc != MPE_OK) return q.rc;
  uint32_t* b1 = ws_array<uint32_t>(ctx, n * 64);
  uint32_t* b2 = ws_array<uint32_t>(ctx, n * 128);
  uint32_t* e2 = ws_array<uint32_t>(ctx, n * 8);
  int32_t* key = ws_array<int32_t>(ctx, n);
  uint32_t* out = ws_array<uint32_t>(ctx, n * 128);
  if (!b1 || !b2 || !e2 || !key || !out) { mpe_set_error_msg("round1: workspace under-reserved"); return MPE_E_NOMEM; }
  auto gather = [&](size_t cnt, Rows src, int sw, int words, uint32_t* dst) {
    if (cnt) hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((cnt * words + 255) / 256)), dim3(256), 0, st, (int)cnt, src, sw, words, dst);
  };
  gather(nVI, s_vi, 64, 64, b1);              gather(nMB, rows(mb_r, 64), 64, 64, b1 + nVI * 64);
  gather(nVI, rows(cinv, 128), 128, 128, b2); gather(nMB, ca_mb, 128, 128, b2 + nVI * 128);
  gather(nVI, e_vi, 8, 8, e2);                gather(nMB, rows(bsel, 8), 8, 8, e2 + nVI * 8);
  (void)hipMemcpyAsync(key, kpub_vi, nVI * 4, hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(key + nVI, kpub_mb, nMB * 4, hipMemcpyDeviceToDevice, st);
  MPE_TRY(modexp_nn2(ctx, pk, (int)n, key_selector(pk, key), rows(b1, 64, nullptr, 64), key_rows(pk, pk->N, 64, key), 64, rows(b2, 128), rows(e2, 8), 8, out, st));
  *m_vi = out; *inv_ok_vi = inv_ok; *x_mb = out + nVI * 128;
  return MPE_OK;
}

// ---- Round1::proceed (rounds.rs:122-206) -----------------------------------------------------------------------------
static int round1(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_off, uint32_t* d_out, hipStream_t st) {
  int rc = round_enter(s, 1, d_in, d_out, true, true);
  if (rc != MPE_OK) return rc;
  mpe_ctx* ctx = s->ctx; const mpe_gg20_keys* K = s->K; const Dim& d = s->d; const Counts c = counts_of(d); const mpe_gg20_nonces& Z = s->Z;
  const int n = d.n, P1 = d.S - 1;
  const Slab in0 = slab_of(s, d_in, h_off, 0);
  (void)hipMemsetAsync(d_out, 0, c.nPI * (size_t)msg_words(d.S, n, 1) * 4, st);
  // m_a_vec[..].c and bc_vec of every party (into_vec_including_me)
  GG_LAUNCH(gather_field_kernel, c.SB * 128, in0, d.S, d.B, n * SUB0, 128, s->ca_all);
  GG_LAUNCH(gather_field_kernel, c.SB * 8, in0, d.S, d.B, n * SUB0 + 128, 8, s->com_all);
  Bump t(s->tmp);
  int32_t* sub0_vi = s->sub0_vi;
  uint8_t* ok_vi = s->ok_vi;
  uint32_t *bsel = t.w(c.nMB * 8), *btq = t.w(c.nMB * 8), *c_b = t.w(c.nMB * 128);
  uint32_t *Bpk = t.w(c.nMB * 16), *BR = t.w(c.nMB * 16), *Bz = t.w(c.nMB * 8), *BTpk = t.w(c.nMB * 16), *BTR = t.w(c.nMB * 16), *BTz = t.w(c.nMB * 8);
  GG_LAUNCH(idx1_kernel, c.nVI, d, in0, sub0_vi);
  // small batches: the verification of the peers' range proofs and the construction of my MessageBs are independent
  // (the reference runs them back to back inside MessageB::b) — two streams, one workspace reservation
  const bool par = ctx->allow_par && (int)c.nVI <= ctx->par_items;
  // the ladders of both halves of the round in ONE launch (round1_merged_ladders): always for large batches, and for a small batch (par)
  // when its two ladder launches on forked streams would overfill the chip (more than 3/4 of the resident groups between them) — then
  // the longer one sets the pace and the merged launch wins although the inversion of c moves in front of it (2 048 sessions:
  // 55.2 || 90.8 ms against 63.5 ms, +9 % on the batch).  Below that the forked launches overlap and merging only serialises the
  // inversion (1 536 sessions: -1.4 %; 1 024: no change; 512: -7.5 %) — profiles/r05/ab_merge_small_batches.jsonl.  What remains of the two halves (the N~ side of the
  // verification, MessageB's encryption tail and DLog proofs) runs on forked streams behind the merged launch.
  const bool merged = round1_merges(ctx, c);
  const bool ahead = merged && s->cinv_ahead;                 // round 0 inverted the ciphertexts already (lock-step signing of a small batch)
  s->cinv_ahead = false;
  if (par && !merged && rc == MPE_OK) { rc = ws_reserve(ctx, ws_need_alice_verify((int)c.nVI) + ws_need_mul_add_enc((int)c.nMB), st); if (rc == MPE_OK) ctx->ws_hold++; }
  bool held = par && !merged && rc == MPE_OK;
  const uint32_t *m_vi = nullptr, *x_mb = nullptr;
  const uint8_t* inv_ok_vi = nullptr;
  AliceProofRows pr{rows(d_in, SUB0, sub0_vi), rows(d_in + 64, SUB0, sub0_vi), rows(d_in + 72, SUB0, sub0_vi),
                    rows(d_in + 136, SUB0, sub0_vi), rows(d_in + 161, SUB0, sub0_vi)};
  if (merged && rc == MPE_OK) {
    rc = ws_reserve(ctx, ws_need_alice_verify((int)c.nVI) + ws_need_mul_add_enc((int)c.nMB) + ws_need_round1_merged(K->pub, c.nVI, c.nMB), st);
    if (rc == MPE_OK) { ctx->ws_hold++; held = true; }
  }
  // small batches: the merged ladder launch goes to the forked stream, in front of MessageB's tail (which needs its x_mb), and the
  // verification's N~ side — fixed-base powers, z^e, an inversion: ~6 ms of short kernels at 1 024 sessions — starts at once on the
  // caller's stream BESIDE it: the ladder's waves are the older ones on their SIMDs and keep 0.92 of their speed (mpe_sched.h), the
  // short kernels run in what is left.  The verification waits for the ladder's output only where it multiplies it in (ev_mid).
  // MessageB's scalars are prepared BEFORE the fork (both branches read them)
  if (rc == MPE_OK && c.nMB > 0)
    hipLaunchKernelGGL(mb_prep_kernel, dim3(blocks_for((int)c.nMB, 64)), dim3(64), 0, st, d, s->gq, s->w, Z.mb_beta_tag, bsel, btq, s->beta);
  Fork g(ctx, st, 2, held && par, 2);
  hipEvent_t m_ready = nullptr;
  auto dlog_proofs = [&](hipStream_t sd) {
    if (rc != MPE_OK || c.nMB == 0) return;
    hipLaunchKernelGGL(dlog_prove_kernel, dim3(blocks_for((int)c.nMB, 64)), dim3(64), 0, sd, (int)c.nMB, ctx->enc, bsel, Z.mb_nonce_b, Bpk, BR, Bz);      // :147
    hipLaunchKernelGGL(dlog_prove_kernel, dim3(blocks_for((int)c.nMB, 64)), dim3(64), 0, sd, (int)c.nMB, ctx->enc, btq, Z.mb_nonce_bt, BTpk, BTR, BTz);   // :148
  };
  // With the inversion done ahead the ladder launch is the first thing the forked stream gets: it has to reach the chip BEFORE the N~ side's
  // kernels take register space on its SIMDs (a SIMD that already holds two of those cannot take a 256-register ladder wave, its unit starts
  // late as the younger wave: +10 ms at 1 024 sessions, +25 ms at 2 048 — profiles/r06/ab_prep18.jsonl).  The two DLog proofs of MessageB
  // (64 - 128 EC waves, ~4 ms, needed only when the round packs its message) go FIRST on the caller's stream: useful work that holds the
  // N~ side back while the ladder's workgroups are placed.  Without the inversion ahead the ladder starts ~7 ms into the round and the
  // same proofs would run into it: they stay behind it.
  const bool dlog_first = ahead && g.on && !ctx->no_r1_dlog_first;
  if (dlog_first) dlog_proofs(st);
  {
    hipStream_t st2 = g.s(1);
    if (merged && rc == MPE_OK) {
      rc = round1_merged_ladders(ctx, K->pub, c.nVI, s->ix.kpub_vi, rows(s->ca_all, 128, s->ix.ca_vi), with_words(pr.s, 64), pr.e, c.nMB, s->ix.kpub_mb,
                                 rows(s->ca_all, 128, s->ix.ca_mb), bsel, Z.mb_r, &m_vi, &inv_ok_vi, &x_mb, st2,
                                 ahead ? s->cinv_pre : nullptr, ahead ? s->cinv_ok_pre : nullptr);
      if (rc == MPE_OK && g.on) { (void)hipEventRecord(ctx->ev_mid, st2); m_ready = ctx->ev_mid; }
      gg_trace(s->ctx, st2, "round 1 merged ladders", rc);
    }
    if (rc == MPE_OK)                                                           // encrypt, Paillier::mul, Paillier::add :133-145
      rc = paillier_mul_add_enc(ctx, K->pub, (int)c.nMB, s->ix.kpub_mb, rows(s->ca_all, 128, s->ix.ca_mb), rows(bsel, 8), 8, Z.mb_beta_tag,
                                Z.mb_r, c_b, st2, x_mb);
    gg_trace(s->ctx, st2, "MessageB ciphertext", rc);
    if (!dlog_first) dlog_proofs(st2);
  }
  if (rc == MPE_OK)        // every range proof of every peer, for both MessageB::b calls (mta/mod.rs:119-131), read in place
    rc = alice_verify(ctx, K->pub, K->stm, (int)c.nVI, s->ix.kpub_vi, s->ix.st_vi, rows(s->ca_all, 128, s->ix.ca_vi), pr, ok_vi, st, m_vi, inv_ok_vi, m_ready);
  gg_trace(s->ctx, st, "alice_verify", rc);
  g.join();
  if (held) ctx->ws_hold--;
  GG_LAUNCH(status1_kernel, c.nPI, d, ok_vi, STAT(1), BADR(1));
  const int per = 2 * P1;
  PACK(c.nMB, per, per, 0, SUB1, 0, c_b, 128); PACK(c.nMB, per, per, 0, SUB1, 128, Bpk, 16); PACK(c.nMB, per, per, 0, SUB1, 144, BR, 16);
  PACK(c.nMB, per, per, 0, SUB1, 160, Bz, 8); PACK(c.nMB, per, per, 0, SUB1, 168, BTpk, 16); PACK(c.nMB, per, per, 0, SUB1, 184, BTR, 16);
  PACK(c.nMB, per, per, 0, SUB1, 200, BTz, 8);
  return round_exit(s, rc, "gg20 round1");
}

// ---- Round2::proceed (rounds.rs:234-317) -----------------------------------------------------------------------------
static int round2(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_off, uint32_t* d_out, hipStream_t st) {
  int rc = round_enter(s, 2, d_in, d_out, true, true);
  if (rc != MPE_OK) return rc;
  mpe_ctx* ctx = s->ctx; const mpe_gg20_keys* K = s->K; const Dim& d = s->d; const Counts c = counts_of(d); const mpe_gg20_nonces& Z = s->Z;
  const Slab in1 = slab_of(s, d_in, h_off, 1);
  (void)hipMemsetAsync(d_out, 0, c.nPI * (size_t)W2 * 4, st);
  GG_LAUNCH(validate_kernel, c.nPI, d, in1, 2, STAT(2), BADR(2));
  Bump t(s->tmp);
  int32_t* sub1_rv = t.i(c.nMB);
  uint32_t *alpha_full = t.w(c.nMB * 64), *alpha = t.w(c.nMB * 8);
  uint8_t* code = t.f(c.nMB);
  Ped ped{s->pedT, t.w(c.nPI * 8), t.w(c.nPI * 16), t.w(c.nPI * 16), t.w(c.nPI * 8), t.w(c.nPI * 8)};
  GG_LAUNCH(idx2_kernel, c.nMB, d, in1, sub1_rv);
  const bool pdl_ahead = rc == MPE_OK && s->lockstep && s->pdl_bn && ctx->use_prio && !ctx->no_pdl_ahead && ctx->use_pair && ctx->use_crt &&
                         ctx->use_pown && ctx->allow_par && (int)c.nPP <= ctx->par_items && c.nPP > 0 && ensure_aux(ctx);
  if (pdl_ahead) {                                              // the auxiliary stream goes on from HERE: behind round 1, beside the decryption
    (void)hipEventRecord(ctx->ev_fork[0], st);
    (void)hipStreamWaitEvent(ctx->aux[1], ctx->ev_fork[0], 0);
  }
  if (rc == MPE_OK)      // Paillier::decrypt of the incoming c_b with my key (mta/mod.rs:165), in place
    rc = paillier_decrypt(ctx, K->prv, (int)c.nMB, s->ix.kown_mb, rows(d_in, SUB1, sub1_rv), alpha_full, st);
  gg_trace(s->ctx, st, "decrypt", rc);
  // (round 6) lock-step signing of a small batch: the two dependent 1024-bit ladders behind beta^N mod N^2 of round 4's PDL proofs
  // (10 ms at 1 024 sessions, on that round's critical path) need nothing but the prover's nonce and key.  They start HERE on an auxiliary
  // stream, at wave priority 0: the decryption ladder (2) and the EC kernels of rounds 2 and 3 (1) keep their speed whichever SIMD they
  // share with them, and the chip — half empty during these 13 ms — does the work.  Round 4 waits for the event, not for the ladders.
  s->pdl_ahead = false;
  if (pdl_ahead && rc == MPE_OK) {
    hipStream_t sa = ctx->aux[1];
    const Rows ksel = sel_of(s->ix.kown_pp, K->prv->nkeys);
    const int keep = ctx->ladder_prio;
    ctx->ladder_prio = 0;
    const int rca = modexp_nn(ctx, K->prv, (int)c.nPP, ksel, rows(Z.pdl_beta, 64, nullptr, 64), tab_rows(K->prv->N, 64, s->ix.kown_pp, K->prv->nkeys), 64,
                              true, s->pdl_bn, sa, true, s->pdl_scratch, s->pdl_phase1 ? 2 : 0);
    ctx->ladder_prio = keep;
    s->pdl_phase1 = false;
    if (rca == MPE_OK) { (void)hipEventRecord(ctx->ev_ahead, sa); s->pdl_ahead = true; } else rc = rca;
    gg_trace(s->ctx, sa, "PDL beta^N, ahead", rc);
  }
  const size_t lanes_fit = ctx->ec_lane_groups ? (size_t)ctx->cus * 4 * 64 * 2 / ctx->device_share : 0;      // two waves per SIMD
  if (c.nMB * 4 <= lanes_fit) GG_LAUNCH(r2a_group_kernel, c.nMB * 4, d, sub1_rv, d_in, alpha_full, s->kq, K->gw, alpha, s->bpk_in, code);
  else GG_LAUNCH(r2a_kernel, c.nMB, d, sub1_rv, d_in, alpha_full, s->kq, K->gw, alpha, s->bpk_in, code);
  GG_LAUNCH(r2b_kernel, c.nPI, d, s->kq, s->gq, s->w, alpha, s->beta, code, Z.l, Z.ped_s1, Z.ped_s2, s->delta_i, s->sigma_i, s->lq, ped,
            STAT(2), BADR(2), alpha_full, s->miu, s->fault_step, s->fault_mask, s->ped_pre);
  PACK(c.nPI, 1, 1, 0, W2, 0, s->delta_i, 8); PACK(c.nPI, 1, 1, 0, W2, 8, ped.T, 16); PACK(c.nPI, 1, 1, 0, W2, 24, ped.e, 8);
  PACK(c.nPI, 1, 1, 0, W2, 32, ped.a1, 16); PACK(c.nPI, 1, 1, 0, W2, 48, ped.a2, 16); PACK(c.nPI, 1, 1, 0, W2, 64, ped.T, 16);
  PACK(c.nPI, 1, 1, 0, W2, 80, ped.z1, 8); PACK(c.nPI, 1, 1, 0, W2, 88, ped.z2, 8);
  return round_exit(s, rc, "gg20 round2");
}

// ---- Round3::proceed (rounds.rs:347-402) -----------------------------------------------------------------------------
static int round3(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_off, uint32_t* d_out, hipStream_t st) {
  int rc = round_enter(s, 3, d_in, d_out, true, true);
  if (rc != MPE_OK) return rc;
  mpe_ctx* ctx = s->ctx; const Dim& d = s->d; const Counts c = counts_of(d);
  const Slab in2 = slab_of(s, d_in, h_off, 2);
  (void)hipMemsetAsync(d_out, 0, c.nPI * (size_t)W3 * 4, st);
  GG_LAUNCH(validate_kernel, c.nPI, d, in2, 3, STAT(3), BADR(3));
  GG_LAUNCH(gather_field_kernel, c.SB * 16, in2, d.S, d.B, 8, 16, s->tvec);                  // t_vec
  const size_t lanes_fit = ctx->ec_lane_groups ? (size_t)ctx->cus * 4 * 64 * 2 / ctx->device_share : 0;
  const int g3 = 2 * d.S <= 4 ? 4 : (2 * d.S <= 8 ? 8 : 16);
  if (c.nPI * g3 <= lanes_fit) GG_LAUNCH(r3_group_kernel, c.nPI * g3, d, g3, in2, s->dinv, STAT(3), BADR(3));
  else GG_LAUNCH(r3_kernel, c.nPI, d, in2, s->dinv, STAT(3), BADR(3));
  PACK(c.nPI, 1, 1, 0, W3, 0, s->Z.blind, 8); PACK(c.nPI, 1, 1, 0, W3, 8, s->g_gamma, 16);          // SignDecommitPhase1
  return round_exit(s, rc, "gg20 round3");
}

// ---- Round4::proceed (rounds.rs:431-498) -----------------------------------------------------------------------------
static int round4(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_off, uint32_t* d_out, hipStream_t st) {
  int rc = round_enter(s, 4, d_in, d_out, true, true);
  if (rc != MPE_OK) return rc;
  mpe_ctx* ctx = s->ctx; const mpe_gg20_keys* K = s->K; const Dim& d = s->d; const Counts c = counts_of(d); const mpe_gg20_nonces& Z = s->Z;
  const int S = d.S, P1 = S - 1;
  const Slab in3 = slab_of(s, d_in, h_off, 3);
  (void)hipMemsetAsync(d_out, 0, c.nPI * (size_t)msg_words(S, d.n, 4) * 4, st);
  GG_LAUNCH(validate_kernel, c.nPI, d, in3, 4, STAT(4), BADR(4));
  // small batches: R and R_dash (two dependent scalar multiplications) beside the Paillier / N~ half of the PDL proofs
  Fork g(ctx, st, 2, ctx->allow_par && (int)c.nPP <= ctx->par_items, 2);
  hipEvent_t R_ready = nullptr;
  if (rc == MPE_OK && c.nPI > 0) {
    hipLaunchKernelGGL(r4_kernel, dim3(blocks_for((int)c.nPI, 64)), dim3(64), 0, g.s(1), d, in3, s->dinv, s->com_all, s->bpk_in, s->kq, s->R, s->Rbar,
                       STAT(4), BADR(4), g.on ? 0 : 1);
    if (g.on) {                       // R is there: u1 of the PDL proofs may start (pdl_prove), R_dash follows on this branch
      (void)hipEventRecord(ctx->ev_mid, g.s(1)); R_ready = ctx->ev_mid;
      hipLaunchKernelGGL(r4_rbar_kernel, dim3(blocks_for((int)c.nPI, 64)), dim3(64), 0, g.s(1), d, s->kq, s->R, s->Rbar);
    }
  }
  Bump t(s->tmp);
  mpe_pdl_proof pp{t.w(c.nPP * 64), t.w(c.nPP * 16), t.w(c.nPP * 128), t.w(c.nPP * 64), t.w(c.nPP * 25), t.w(c.nPP * 64), t.w(c.nPP * 89)};
  mpe_pdl_nonces pn{Z.pdl_alpha, Z.pdl_beta, Z.pdl_rho, Z.pdl_gamma};
  if (rc == MPE_OK)                                                                                            // phase5_proof_pdl
    rc = pdl_prove(ctx, K->prv, K->stm, (int)c.nPP, s->ix.kown_pp, s->ix.st_pp, rows(s->c_a, 128, s->ix.pi_pp), rows(s->Rbar, 16, s->ix.pi_pp),
                   rows(s->R, 16, s->ix.pi_pp), rows(s->kq, 8, s->ix.pi_pp), rows(Z.r_a, 64, s->ix.pi_pp), &pn, &pp, st, &g,
                   s->pdl_ahead ? s->pdl_bn : nullptr, s->pdl_ahead ? ctx->ev_ahead : nullptr, R_ready);
  else g.join();
  if (rc == MPE_OK) s->pdl_ahead = false;                    // consumed (otherwise session_release waits for it)
  gg_trace(s->ctx, st, "pdl_prove", rc);
  PACK(c.nPP, P1, S, 0, SUB4, 0, pp.z, 64); PACK(c.nPP, P1, S, 0, SUB4, 64, pp.u1, 16); PACK(c.nPP, P1, S, 0, SUB4, 80, pp.u2, 128);
  PACK(c.nPP, P1, S, 0, SUB4, 208, pp.u3, 64); PACK(c.nPP, P1, S, 0, SUB4, 272, pp.s1, 25); PACK(c.nPP, P1, S, 0, SUB4, 297, pp.s2, 64);
  PACK(c.nPP, P1, S, 0, SUB4, 361, pp.s3, 89);
  PACK(c.nPI, 1, S, P1, SUB4, 0, s->Rbar, 16);
  return round_exit(s, rc, "gg20 round4");
}

// ---- Round5::proceed (rounds.rs:525-601) -----------------------------------------------------------------------------
static int round5(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_off, uint32_t* d_out, hipStream_t st) {
  int rc = round_enter(s, 5, d_in, d_out, true, true);
  if (rc != MPE_OK) return rc;
  mpe_ctx* ctx = s->ctx; const mpe_gg20_keys* K = s->K; const Dim& d = s->d; const Counts c = counts_of(d); const mpe_gg20_nonces& Z = s->Z;
  const Slab in4 = slab_of(s, d_in, h_off, 4);
  (void)hipMemsetAsync(d_out, 0, c.nPI * (size_t)W5 * 4, st);
  GG_LAUNCH(validate_kernel, c.nPI, d, in4, 5, STAT(5), BADR(5));
  Bump t(s->tmp);
  int32_t *sub4_pv = s->sub4_pv, *rdash_pv = s->rdash_pv;
  uint8_t* ok_pv = s->ok_pv;
  Heg heg{t.w(c.nPI * 16), t.w(c.nPI * 16), t.w(c.nPI * 16), t.w(c.nPI * 8), t.w(c.nPI * 8)};
  GG_LAUNCH(idx5_kernel, c.nPV, d, in4, sub4_pv, rdash_pv);
  const bool par = ctx->allow_par && (int)c.nPV <= ctx->par_items;
  PdlProofRows pr{rows(d_in, SUB4, sub4_pv), rows(d_in + 64, SUB4, sub4_pv), rows(d_in + 80, SUB4, sub4_pv), rows(d_in + 208, SUB4, sub4_pv),
                  rows(d_in + 272, SUB4, sub4_pv), rows(d_in + 297, SUB4, sub4_pv), rows(d_in + 361, SUB4, sub4_pv)};
  Fork g(ctx, st, 2, par, 2);
  if (rc == MPE_OK && c.nPI > 0)
    hipLaunchKernelGGL(r5_prove_kernel, dim3(blocks_for((int)c.nPI, 64)), dim3(64), 0, g.s(1), d, s->R, s->sigma_i, s->lq, s->pedT, Z.heg_s1, Z.heg_s2, heg);
  if (rc == MPE_OK)      // phase5_verify_pdl for every prover (mine included), G = the VERIFIER's R (rounds.rs:546-558)
    rc = pdl_verify(ctx, K->pub, K->stm, (int)c.nPV, s->ix.kpub_pv, s->ix.st_pv, rows(s->ca_all, 128, s->ix.ca_pv), rows(d_in, SUB4, rdash_pv),
                    rows(s->R, 16, s->ix.pi_pv), pr, ok_pv, st);
  gg_trace(s->ctx, st, "pdl_verify", rc);
  g.join();
  GG_LAUNCH(r5_status_kernel, c.nPI, d, in4, ok_pv, STAT(5), BADR(5));
  PACK(c.nPI, 1, 1, 0, W5, 0, heg.S, 16); PACK(c.nPI, 1, 1, 0, W5, 16, heg.T, 16); PACK(c.nPI, 1, 1, 0, W5, 32, heg.A3, 16);
  PACK(c.nPI, 1, 1, 0, W5, 48, heg.z1, 8); PACK(c.nPI, 1, 1, 0, W5, 56, heg.z2, 8);
  return round_exit(s, rc, "gg20 round5");
}

// ---- Round6::proceed (rounds.rs:612-636) -----------------------------------------------------------------------------
static int round6(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_off, hipStream_t st) {
  int rc = round_enter(s, 6, d_in, nullptr, true, false);
  if (rc != MPE_OK) return rc;
  mpe_ctx* ctx = s->ctx; const Dim& d = s->d; const Counts c = counts_of(d);
  const Slab in5 = slab_of(s, d_in, h_off, 5);
  GG_LAUNCH(validate_kernel, c.nPI, d, in5, 6, STAT(6), BADR(6));
  const size_t lanes_fit = ctx->ec_lane_groups ? (size_t)ctx->cus * 4 * 64 * 2 / ctx->device_share : 0;
  const int g6 = 3 * d.S <= 8 ? 8 : (3 * d.S <= 16 ? 16 : 32);
  if (c.nPI * g6 <= lanes_fit) GG_LAUNCH(r6_group_kernel, c.nPI * g6, d, g6, in5, s->R, s->tvec, s->K->y, STAT(6), BADR(6));
  else GG_LAUNCH(r6_kernel, c.nPI, d, in5, s->R, s->tvec, s->K->y, STAT(6), BADR(6));
  return round_exit(s, rc, "gg20 round6");
}

// ---- Round7::new (rounds.rs:672-692) -> PartialSignature ----------------------------------------------------------------
static int round7(mpe_gg20_session* s, const uint32_t* d_msg, uint32_t* d_out, hipStream_t st) {
  int rc = round_enter(s, 7, d_msg, d_out, true, true);
  if (rc != MPE_OK) return rc;
  const Dim& d = s->d; const Counts c = counts_of(d);
  GG_LAUNCH(r7_kernel, c.nPI, d, d_msg, s->R, s->kq, s->sigma_i, s->mq, s->rq, s->s_i, s->fault_step, s->fault_mask);
  PACK(c.nPI, 1, 1, 0, W6, 0, s->s_i, 8);
  return round_exit(s, rc, "gg20 round7");
}
// ---- SignManual::complete (sign.rs:625-646) ---------------------------------------------------------------------------------
static int complete(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_off, hipStream_t st) {
  int rc = round_enter(s, 8, d_in, nullptr, true, false);
  if (rc != MPE_OK) return rc;
  const Dim& d = s->d; const Counts c = counts_of(d);
  const Slab in6 = slab_of(s, d_in, h_off, 7);
  GG_LAUNCH(complete_kernel, c.nPI, d, in6, s->R, s->mq, s->rq, s->s_i, s->K->y, s->status, s->bad, s->sig_r, s->sig_s, s->sig_recid);
  return round_exit(s, rc, "gg20 complete");
}

static void session_release(mpe_gg20_session* s, hipStream_t st) {
  if (!s) return;
  if (s->pdl_ahead && s->ctx->ev_ahead) { (void)hipStreamWaitEvent(st, s->ctx->ev_ahead, 0); s->pdl_ahead = false; }    // started, never consumed (a failed round)
  if (s->pdl_phase1 && s->ctx->ev_ahead && s->ctx->aux_ready) {                                                            // its first ladder may still run
    (void)hipEventRecord(s->ctx->ev_ahead, s->ctx->aux[1]);
    (void)hipStreamWaitEvent(st, s->ctx->ev_ahead, 0);
    s->pdl_phase1 = false;
  }
  if (s->mem) {
    // secrets (k_i, gamma_i, w_i, sigma_i, nonce-derived intermediates) do not outlive the object (range_proofs.rs:26-36 zeroizes)
    (void)hipMemsetAsync(s->mem, 0, s->mem_bytes, st);
    if (s->from_cache) s->ctx->sess_in_use = false;
    else { (void)hipStreamSynchronize(st); (void)hipFree(s->mem); }
  }
  delete s;
}

}  // namespace gg
}  // namespace mpe

extern "C" {

int mpe_gg20_keys_create(mpe_ctx* ctx, int t, int n, int n_signers, const int32_t* h_signers, int nkeysets, int n_own,
                         const int32_t* h_own, const uint32_t* d_x, const uint32_t* d_p, const uint32_t* d_q, const uint32_t* d_N,
                         const uint32_t* d_Nt, const uint32_t* d_h1, const uint32_t* d_h2, const uint32_t* d_y, const uint32_t* d_X,
                         mpe_gg20_keys** out, void* stream) {
  if (!ctx || !h_signers || !h_own || !d_x || !d_p || !d_q || !d_N || !d_Nt || !d_h1 || !d_h2 || !d_y || !d_X || !out) return MPE_E_ARG;
  if (n < 2 || n > 8 || n_signers < 2 || n_signers > n || t < 1 || n_signers <= t || nkeysets < 1 || n_own < 1 || n_own > n) return MPE_E_ARG;
  for (int i = 0; i < n_signers; ++i)
    if (h_signers[i] < 0 || h_signers[i] >= n || (i && h_signers[i] <= h_signers[i - 1])) return MPE_E_ARG;
  for (int i = 0; i < n_own; ++i)
    if (h_own[i] < 0 || h_own[i] >= n || (i && h_own[i] <= h_own[i - 1])) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  mpe_gg20_keys* K = new (std::nothrow) mpe_gg20_keys();
  if (!K) return MPE_E_NOMEM;
  K->t = t; K->n = n; K->S = n_signers; K->K = nkeysets; K->n_own = n_own;
  for (int a = 0; a < 8; ++a) K->own_slot[a] = -1;
  for (int i = 0; i < n_signers; ++i) K->signers[i] = h_signers[i];
  for (int i = 0; i < n_own; ++i) { K->own[i] = h_own[i]; K->own_slot[h_own[i]] = i; }
  const size_t kk = (size_t)nkeysets;
  const size_t words = kk * n_own * 8 + kk * n * 16 + kk * 16 + kk * n_signers * 16 + 32 + kk * n * (64 + 72 + 88);
  hipError_t e = hipMalloc(&K->blob, words * 4);
  if (e != hipSuccess) { delete K; mpe_set_error("hipMalloc(gg20 keys)", e); return MPE_E_NOMEM; }
  K->blob_bytes = words * 4;
  K->x = (uint32_t*)K->blob; K->X = K->x + kk * n_own * 8; K->y = K->X + kk * n * 16; K->gw = K->y + kk * 16;
  K->bounds.q = K->gw + kk * n_signers * 16; K->bounds.q3 = K->bounds.q + 8; K->bounds.Nm2 = K->bounds.q3 + 24;
  K->bounds.qNt = K->bounds.Nm2 + kk * n * 64; K->bounds.q3Nt = K->bounds.qNt + kk * n * 72;
  hipLaunchKernelGGL(mpe::smp::bounds_kernel, dim3(mpe::blocks_for(nkeysets * n + 1, 64)), dim3(64), 0, st, nkeysets * n, d_N, d_Nt, K->bounds);
  (void)hipMemcpyAsync(K->x, d_x, kk * n_own * 8 * 4, hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(K->X, d_X, kk * n * 16 * 4, hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(K->y, d_y, kk * 16 * 4, hipMemcpyDeviceToDevice, st);
  {
    mpe::gg::Dim d{};
    d.S = n_signers; d.n = n; d.K = nkeysets;
    for (int i = 0; i < n_signers; ++i) d.sg[i] = h_signers[i];
    hipLaunchKernelGGL(mpe::gg::gw_kernel, dim3(mpe::blocks_for(nkeysets * n_signers, 64)), dim3(64), 0, st, d, K->X, K->gw);
  }
  int rc = mpe_paillier_create_public(ctx, nkeysets * n, d_N, &K->pub, stream);
  if (rc == MPE_OK) rc = mpe_paillier_create_private(ctx, nkeysets * n_own, d_p, d_q, &K->prv, stream);
  if (rc == MPE_OK) {
    // window width of the fixed-base tables of h1, h2: the widest whose tables (2 K n bases) stay within the budget —
    // 13 bits (0.5 GB per base) for a handful of key sets, narrower when a batch carries many wallets
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t budget = ctx->fb_budget_bytes ? ctx->fb_budget_bytes : free_b / 4;
    // thousands of wallets: rather half of the free memory for 4-bit tables (704 multiplications per power) than none at all (a
    // 2816-bit variable-base ladder: 4.8x the multiplications) — K = 4 096 wallets x 3 parties: 80 GB
    if (!ctx->fb_budget_bytes && mpe_statements_table_bytes(nkeysets * n, 4) > budget && mpe_statements_table_bytes(nkeysets * n, 4) <= free_b / 2) budget = free_b / 2;
    int wb = ctx->fb_window_bits;
    while (wb > 4 && mpe_statements_table_bytes(nkeysets * n, wb) > budget) --wb;
    // thousands of wallets with moduli of their own: even 4-bit tables (3.2 MB per base) do not fit — no tables at all, the powers of
    // h1, h2 run on the variable-base ladder modulo N~ (mpe_statements_create_wb(wb = 0)); mpe_gg20_keys_fb_window_bits then says 0
    if (mpe_statements_table_bytes(nkeysets * n, wb) > budget) wb = 0;
    rc = mpe_statements_create_wb(ctx, nkeysets * n, d_Nt, d_h1, d_h2, wb, &K->stm, stream);
  }
  if (rc != MPE_OK) { mpe_gg20_keys_destroy(K); return rc; }
  *out = K;
  return MPE_OK;
}

int mpe_gg20_keys_destroy(mpe_gg20_keys* K) {
  if (!K) return MPE_E_ARG;
  if (K->pub) mpe_paillier_destroy(K->pub);
  if (K->prv) mpe_paillier_destroy(K->prv);
  if (K->stm) mpe_statements_destroy(K->stm);
  if (K->blob) { (void)hipMemset(K->blob, 0, K->blob_bytes); (void)hipFree(K->blob); }      // the key shares
  delete K;
  return MPE_OK;
}
int mpe_gg20_keys_fb_window_bits(const mpe_gg20_keys* K) { return (K && K->stm) ? K->stm->fb_wb : MPE_E_ARG; }

int mpe_gg20_msg_words(int n_signers, int n, int round) { return mpe::gg::msg_words(n_signers, n, round); }

int mpe_gg20_session_create(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, int n_local, const int32_t* h_local,
                            const int32_t* d_keyset, const mpe_gg20_nonces* nonces, int dedup_verify, mpe_gg20_session** out, void* stream) {
  if (!ctx || !keys || !nonces || !out || batch <= 0 || n_local < 1 || n_local > keys->S || !h_local) return MPE_E_ARG;
  for (int i = 0; i < n_local; ++i) {
    if (h_local[i] < 0 || h_local[i] >= keys->S || (i && h_local[i] <= h_local[i - 1])) return MPE_E_ARG;
    if (keys->own_slot[keys->signers[h_local[i]]] < 0) { mpe_set_error_msg("gg20: a local party's secrets are not in the key object"); return MPE_E_ARG; }
  }
  if (keys->K > 1 && !d_keyset) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  mpe_gg20_session* s = new (std::nothrow) mpe_gg20_session();
  if (!s) return MPE_E_NOMEM;
  s->ctx = ctx; s->K = keys; s->B = batch; s->L = n_local; s->dedup = dedup_verify ? 1 : 0; s->Z = *nonces;
  mpe::gg::Dim& d = s->d;
  d.B = batch; d.S = keys->S; d.n = keys->n; d.L = n_local; d.V = dedup_verify ? 1 : 2; d.PV = dedup_verify ? 1 : n_local; d.K = keys->K;
  d.n_own = keys->n_own; d.ks = d_keyset; d.enc = ctx->enc;
  for (int i = 0; i < 8; ++i) { d.loc[i] = i < n_local ? h_local[i] : 0; d.sg[i] = keys->signers[i]; d.oslot[i] = keys->own_slot[i] < 0 ? 0 : keys->own_slot[i]; }
  const size_t bytes = mpe::gg::layout(s, nullptr);
  if (!ctx->sess_in_use) {
    if (bytes > ctx->sess_bytes) {
      if (ctx->sess_buf) { (void)hipStreamSynchronize(st); (void)hipFree(ctx->sess_buf); ctx->sess_buf = nullptr; ctx->sess_bytes = 0; }
      const hipError_t e = hipMalloc(&ctx->sess_buf, bytes);
      if (e != hipSuccess) { delete s; mpe_set_error("hipMalloc(gg20 session)", e); return MPE_E_NOMEM; }
      ctx->sess_bytes = bytes;
    }
    s->mem = ctx->sess_buf; s->from_cache = true; ctx->sess_in_use = true;
  } else {
    const hipError_t e = hipMalloc(&s->mem, bytes);
    if (e != hipSuccess) { delete s; mpe_set_error("hipMalloc(gg20 session)", e); return MPE_E_NOMEM; }
  }
  s->mem_bytes = bytes;
  (void)mpe::gg::layout(s, (char*)s->mem);
  const mpe::gg::Counts c = mpe::gg::counts_of(d);
  (void)hipMemsetAsync(s->status, 0, c.nPI * mpe::gg::NR * 4, st);
  (void)hipMemsetAsync(s->bad, 0, c.nPI * mpe::gg::NR * 4, st);
  size_t total = c.nVI;
  if (c.nMB > total) total = c.nMB;
  if (c.nAP > total) total = c.nAP;
  if (c.nPV > total) total = c.nPV;
  if (c.nPI > total) total = c.nPI;
  hipLaunchKernelGGL(mpe::gg::idx_kernel, dim3(mpe::blocks_for((int)total, 64)), dim3(64), 0, st, d, s->ix, (int)total);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("gg20 idx_kernel", e); mpe::gg::session_release(s, st); return MPE_E_HIP; }
  *out = s;
  return MPE_OK;
}
int mpe_gg20_session_destroy(mpe_gg20_session* s, void* stream) {
  if (!s) return MPE_E_ARG;
  mpe::gg::session_release(s, (hipStream_t)stream);
  return MPE_OK;
}

int mpe_gg20_session_rearm(mpe_gg20_session* s, const int32_t* d_keyset, const mpe_gg20_nonces* nonces, void* stream) {
  // a long-lived party process signs batch after batch with the same (keys, batch, local parties) shape: the state arrays and
  // the index tables stay, everything nonce-derived is zeroed and the round counter starts again
  if (!s || !nonces) return MPE_E_ARG;
  if (s->K->K > 1 && !d_keyset) return MPE_E_ARG;
  // a batch that is half-way through the protocol is not silently thrown away: finish it (round 8 = complete) or destroy it
  // — unless a round call FAILED (MPE_E_NOMEM, a HIP error: the batch cannot go on) or the caller abandoned it (mpe_gg20_session_abort)
  if (s->next_round != 0 && s->next_round <= 8 && !s->failed) { mpe_set_error_msg("gg20 session rearm: the previous batch has not completed (finish it, or mpe_gg20_session_abort)"); return MPE_E_ARG; }
  // The caller passes FRESHLY SAMPLED values for every batch (re-using k, gamma or a Paillier randomness leaks the key share);
  // the library cannot tell fresh from stale device arrays — the arrays may legitimately be the same buffers refilled.
  hipStream_t st = (hipStream_t)stream;
  const mpe::gg::Counts c = mpe::gg::counts_of(s->d);
  // the state region after the index tables: secrets of the previous batch do not outlive it
  const hipError_t em = hipMemsetAsync(s->kq, 0, (size_t)((char*)s->mem + s->mem_bytes - (char*)s->kq), st);
  if (em != hipSuccess) { mpe_set_error("gg20 session rearm (wipe)", em); return MPE_E_HIP; }
  s->Z = *nonces; s->d.ks = d_keyset; s->next_round = 0; s->failed = false; s->fault_step = 0; s->fault_mask = 0;
  size_t total = c.nVI;                 // the key indices follow the (possibly different) key-set choice
  if (c.nMB > total) total = c.nMB;
  if (c.nAP > total) total = c.nAP;
  if (c.nPV > total) total = c.nPV;
  if (c.nPI > total) total = c.nPI;
  hipLaunchKernelGGL(mpe::gg::idx_kernel, dim3(mpe::blocks_for((int)total, 64)), dim3(64), 0, st, s->d, s->ix, (int)total);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("gg20 session rearm", e); return MPE_E_HIP; }
  return MPE_OK;
}

int mpe_gg20_session_abort(mpe_gg20_session* s, void* stream) {
  // the caller gives the running batch up (it stopped after the offline stage, a peer vanished, a round call failed): everything
  // nonce-derived is wiped now and the object waits for mpe_gg20_session_rearm; the round entry points refuse until then
  if (!s) return MPE_E_ARG;
  const hipError_t em = hipMemsetAsync(s->kq, 0, (size_t)((char*)s->mem + s->mem_bytes - (char*)s->kq), (hipStream_t)stream);
  if (em != hipSuccess) { mpe_set_error("gg20 session abort (wipe)", em); return MPE_E_HIP; }
  s->failed = true;
  s->next_round = 9;
  return MPE_OK;
}

int mpe_gg20_session_fault_inject(mpe_gg20_session* s, int step, uint32_t party_mask) {
  if (!s || (step != 0 && step != 5 && step != 6 && step != 7)) return MPE_E_ARG;
  s->fault_step = step; s->fault_mask = party_mask;
  return MPE_OK;
}
int mpe_gg20_round0(mpe_gg20_session* s, uint32_t* d_out, void* stream) { return s ? mpe::gg::round0(s, d_out, (hipStream_t)stream) : MPE_E_ARG; }
int mpe_gg20_round1(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream) {
  return s ? mpe::gg::round1(s, d_in, h_in_off, d_out, (hipStream_t)stream) : MPE_E_ARG;
}
int mpe_gg20_round2(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream) {
  return s ? mpe::gg::round2(s, d_in, h_in_off, d_out, (hipStream_t)stream) : MPE_E_ARG;
}
int mpe_gg20_round3(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream) {
  return s ? mpe::gg::round3(s, d_in, h_in_off, d_out, (hipStream_t)stream) : MPE_E_ARG;
}
int mpe_gg20_round4(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream) {
  return s ? mpe::gg::round4(s, d_in, h_in_off, d_out, (hipStream_t)stream) : MPE_E_ARG;
}
int mpe_gg20_round5(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_in_off, uint32_t* d_out, void* stream) {
  return s ? mpe::gg::round5(s, d_in, h_in_off, d_out, (hipStream_t)stream) : MPE_E_ARG;
}
int mpe_gg20_round6(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_in_off, void* stream) {
  return s ? mpe::gg::round6(s, d_in, h_in_off, (hipStream_t)stream) : MPE_E_ARG;
}
int mpe_gg20_round7(mpe_gg20_session* s, const uint32_t* d_msg, uint32_t* d_out, void* stream) {
  return s ? mpe::gg::round7(s, d_msg, d_out, (hipStream_t)stream) : MPE_E_ARG;
}
int mpe_gg20_complete(mpe_gg20_session* s, const uint32_t* d_in, const int64_t* h_in_off, void* stream) {
  return s ? mpe::gg::complete(s, d_in, h_in_off, (hipStream_t)stream) : MPE_E_ARG;
}
int mpe_gg20_session_result(const mpe_gg20_session* s, int32_t* d_status, uint32_t* d_bad_actors, uint32_t* d_r, uint32_t* d_s,
                            int32_t* d_recid, uint32_t* d_R, void* stream) {
  if (!s) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const mpe::gg::Counts c = mpe::gg::counts_of(s->d);
  hipLaunchKernelGGL(mpe::gg::result_kernel, dim3(mpe::blocks_for((int)c.nPI, 64)), dim3(64), 0, st, s->d, s->status, s->bad, s->R, d_status,
                     d_bad_actors, d_R);
  if (s->next_round > 8) {
    if (d_r) (void)hipMemcpyAsync(d_r, s->sig_r, c.nPI * 32, hipMemcpyDeviceToDevice, st);
    if (d_s) (void)hipMemcpyAsync(d_s, s->sig_s, c.nPI * 32, hipMemcpyDeviceToDevice, st);
    if (d_recid) (void)hipMemcpyAsync(d_recid, s->sig_recid, c.nPI * 4, hipMemcpyDeviceToDevice, st);
  } else if (d_r || d_s || d_recid) {
    mpe_set_error_msg("gg20: the signature exists after mpe_gg20_complete");
    return MPE_E_ARG;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("gg20 result", e); return MPE_E_HIP; }
  return MPE_OK;
}

int mpe_gg20_sign(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, const int32_t* d_keyset, const mpe_gg20_nonces* nonces, uint32_t* d_r,
                  uint32_t* d_s, int32_t* d_recid, uint32_t* d_R, int32_t* d_status, int dedup_verify, int chunk, void* stream) {
  if (!ctx || !keys || !nonces || !d_r || !d_s || !d_recid || !d_status || batch < 0) return MPE_E_ARG;
  const int S = keys->S, n = keys->n;
  for (int i = 0; i < S; ++i) if (keys->own_slot[keys->signers[i]] < 0) { mpe_set_error_msg("mpe_gg20_sign needs every signer's secrets"); return MPE_E_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (chunk <= 0) {
    // 65 536 sessions per pass (the EC kernels want >= 2 waves per SIMD), fewer when the shape is wide: state, message
    // slabs and composite workspace per session grow like S (S-1) n; keep a pass under ~64 GB
    const size_t P = (size_t)S * (S - 1), V = dedup_verify ? 1 : 2, PV = dedup_verify ? 1 : S;
    const size_t nVI = P * V * n, nPV = PV * P;
    int mw = 0;
    for (int r = 0; r < 8; ++r) { const int w = mpe::gg::msg_words(S, n, r); if (w > mw) mw = w; }
    const size_t words = (size_t)S * 1200 + (size_t)S * n * 260 + nVI * 8 + P * 2 * 300 + P * 470 + nPV * 8 +      // state + round scratch
                         (nVI > nPV ? nVI : nPV) * 2300 + (size_t)S * n * 2100 +                                     // composite workspace
                         2 * (size_t)S * mw;                                                                         // two message slabs
    size_t fit = ((size_t)64 << 30) / (words * 4);
    fit = fit >= 1024 ? (fit / 1024) * 1024 : (fit ? fit : 1);
    chunk = (int)(fit < 65536 ? fit : 65536);
  }
  int32_t local[8];
  for (int i = 0; i < S; ++i) local[i] = i;
  int maxw = 0;
  for (int r = 0; r < 8; ++r) { const int w = mpe::gg::msg_words(S, n, r); if (w > maxw) maxw = w; }
  const int Bc = batch < chunk ? batch : chunk;
  // two message slabs alternate: round q writes slab q & 1 and reads the other
  size_t slab_off[7];
  const size_t slab_one = (size_t)S * Bc * maxw, slab_words = 2 * slab_one;
  for (int q = 0; q < 7; ++q) slab_off[q] = (q & 1) ? slab_one : 0;
  const size_t res_words = (size_t)S * Bc * (1 + 8 + 8 + 1 + 16);
  const size_t need = (slab_words + res_words) * 4 + 4096;
  if (need > ctx->slab_bytes) {
    if (ctx->slab_buf) { (void)hipStreamSynchronize(st); (void)hipFree(ctx->slab_buf); ctx->slab_buf = nullptr; ctx->slab_bytes = 0; }
    const hipError_t e = hipMalloc(&ctx->slab_buf, need);
    if (e != hipSuccess) { mpe_set_error("hipMalloc(gg20 message slabs)", e); return MPE_E_NOMEM; }
    ctx->slab_bytes = need;
  }
  uint32_t* M[7];
  for (int q = 0; q < 7; ++q) M[q] = (uint32_t*)ctx->slab_buf + slab_off[q];
  int32_t* pst = (int32_t*)((uint32_t*)ctx->slab_buf + slab_words);
  uint32_t* pr = (uint32_t*)(pst + (size_t)S * Bc);
  uint32_t* ps = pr + (size_t)S * Bc * 8;
  int32_t* prec = (int32_t*)(ps + (size_t)S * Bc * 8);
  uint32_t* pR = (uint32_t*)(prec + (size_t)S * Bc);
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int B = batch - b0 < chunk ? batch - b0 : chunk;
    const size_t oPI = (size_t)b0 * S, oAP = oPI * n, oPP = oPI * (S - 1), oMB = oPP * 2;
    mpe_gg20_nonces Z = *nonces;
    Z.k += oPI * 8; Z.gamma += oPI * 8; Z.blind += oPI * 8; Z.r_a += oPI * 64;
    Z.al_alpha += oAP * 24; Z.al_beta += oAP * 64; Z.al_gamma += oAP * 88; Z.al_rho += oAP * 72;
    Z.mb_beta_tag += oMB * 64; Z.mb_r += oMB * 64; Z.mb_nonce_b += oMB * 8; Z.mb_nonce_bt += oMB * 8;
    Z.l += oPI * 8; Z.ped_s1 += oPI * 8; Z.ped_s2 += oPI * 8;
    Z.pdl_alpha += oPP * 24; Z.pdl_beta += oPP * 64; Z.pdl_rho += oPP * 72; Z.pdl_gamma += oPP * 88;
    Z.heg_s1 += oPI * 8; Z.heg_s2 += oPI * 8; Z.msg += (size_t)b0 * 8;
    mpe_gg20_session* s = nullptr;
    int rc = mpe_gg20_session_create(ctx, keys, B, S, local, d_keyset ? d_keyset + b0 : nullptr, &Z, dedup_verify, &s, stream);
    if (rc != MPE_OK) return rc;
    s->lockstep = true;                           // the message slabs never leave the library between two rounds
    rc = mpe::gg::round0(s, M[0], st);
    if (rc == MPE_OK) rc = mpe::gg::round1(s, M[0], nullptr, M[1], st);
    if (rc == MPE_OK) rc = mpe::gg::round2(s, M[1], nullptr, M[2], st);
    if (rc == MPE_OK) rc = mpe::gg::round3(s, M[2], nullptr, M[3], st);
    if (rc == MPE_OK) rc = mpe::gg::round4(s, M[3], nullptr, M[4], st);
    if (rc == MPE_OK) rc = mpe::gg::round5(s, M[4], nullptr, M[5], st);
    if (rc == MPE_OK) rc = mpe::gg::round6(s, M[5], nullptr, st);
    if (rc == MPE_OK) rc = mpe::gg::round7(s, Z.msg, M[6], st);
    if (rc == MPE_OK) rc = mpe::gg::complete(s, M[6], nullptr, st);
    if (rc == MPE_OK) rc = mpe_gg20_session_result(s, pst, nullptr, pr, ps, prec, pR, stream);
    if (rc == MPE_OK)
      hipLaunchKernelGGL(mpe::gg::sign_finish_kernel, dim3(mpe::blocks_for(B, 64)), dim3(64), 0, st, B, S, b0, pst, pr, ps, prec, pR, d_r, d_s,
                         d_recid, d_R, d_status);
    mpe::gg::session_release(s, st);
    if (rc != MPE_OK) return rc;
  }
  // the message slabs and the composite workspace held nonce-derived values of this call
  (void)hipMemsetAsync(ctx->slab_buf, 0, need, st);
  return mpe_ctx_wipe(ctx, stream);
}

}  // extern "C"
