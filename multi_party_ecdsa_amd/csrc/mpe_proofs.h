// secp256k1 batch ops, modular inversion, and the zero-knowledge proofs of the GG20 hot path:
//   AliceProof::{generate,verify}          src/utilities/mta/range_proofs.rs:39-193
//   PDLwSlackProof::{prove,verify}         src/utilities/zk_pdl_with_slack/mod.rs:68-199
//   DLogProof::{prove,verify}              curv, used at src/utilities/mta/mod.rs:147-148,170-171
// Included by mpe_lib.hip.  Heavy arithmetic = launch_modexp / launch_modmul; everything else is
// one-item-per-lane glue (mpe_small.h, mpe_ec.h).
#pragma once
#include "mpe_ec.h"
#include "mpe_internal.h"
#include "mpe_modinv.h"
#include "mpe_paillier.h"
#include "mpe_small.h"

struct mpe_statements {
  int count = 0;
  void* blob = nullptr;
  uint32_t* Nt = nullptr;   // [count][64]
  uint32_t* h1 = nullptr;
  uint32_t* h2 = nullptr;
  mpe_modset* ms = nullptr;  // 2048-bit, modulus k = N~_k
  int fb_wb = 8;               // window width of the fixed-base tables
  uint32_t* fb_tab = nullptr;  // [2*count][windows][2^fb_wb][72] fixed-base tables of h1 (even) / h2 (odd)
};

#include "mpe_fixedbase.h"

// internal: statement set with an explicit window width of its fixed-base tables (the GG20 key object sizes it by memory)
extern "C" int mpe_statements_create_wb(mpe_ctx* ctx, int count, const uint32_t* d_Nt, const uint32_t* d_h1, const uint32_t* d_h2,
                                        int wb, mpe_statements** out, void* stream);
// bytes of the fixed-base tables of `count` statements at window width wb
static inline size_t mpe_statements_table_bytes(int count, int wb) {
  return (size_t)2 * count * mpe::fb_windows(wb) * ((size_t)1 << wb) * mpe::Cfg2048::K * sizeof(uint32_t);
}

namespace mpe {

static int launch_fb_modexp(mpe_ctx* ctx, const mpe_statements* stm, int B, Rows st_sel, int which, Rows exps, int ew,
                            uint32_t* out, hipStream_t st) {
  if (B == 0) return MPE_OK;
  using C = Cfg2048;
  const int cap = ctx->cus * ctx->modexp_waves_per_cu;
  // lane groups per item (mpe_fixedbase.h): the widest split that still leaves one wave per SIMD, so that the chain of
  // multiplications — which is all a small launch's time — is as short as the chip's idle lanes allow; 1 for large launches
  int split = 1;
  if (ctx->fb_split > 0) { while (split * 2 <= ctx->fb_split && split * 2 <= (int)C::GROUPS) split *= 2; }
  else if (ctx->adaptive_lanes) { while (split * 2 <= (int)C::GROUPS && ((long)B * split * 2 + C::GROUPS - 1) / C::GROUPS <= cap / 2) split *= 2; }
  const int per_wave = C::GROUPS / split;
  const int units = (B + per_wave - 1) / per_wave;
  const int grid = ladder_grid(ctx, units, cap);
  int32_t* sst = (int32_t*)tables_for(ctx, SCHED_WORDS * sizeof(int32_t), st);       // the scheduler's state lives in the stream slot's scratch
  if (!sst) return MPE_E_NOMEM;
  const SchedArgs sched = ladder_sched(ctx, units, cap, sst, st);
  ModsetView v;
  v.n_limbs = stm->ms->n_limbs; v.one_limbs = stm->ms->one_limbs; v.r2_limbs = stm->ms->r2_limbs;
  v.r2h_limbs = stm->ms->r2h_limbs; v.n0inv = stm->ms->n0inv; v.count = stm->ms->count;
  prof_begin(ctx, st, 5, C::BITS, ew, B, stm->fb_wb);          // kind 5: fixed-base ladder; exp2_words carries the window width
  hipLaunchKernelGGL(fb_modexp_kernel<C>, dim3(grid), dim3(64), 0, st, B, v, st_sel, which, stm->fb_tab, stm->fb_wb, exps, ew, out, sched, split);
  prof_end(ctx, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("fb_modexp_kernel", e); return MPE_E_HIP; }
  return MPE_OK;
}

// ---------------------------------------------------------------------------------------------
// secp256k1 batch kernels
// ---------------------------------------------------------------------------------------------
// out = (k mod q) * P;  P == nullptr -> generator
__global__ void __launch_bounds__(64) MPE_EC_OCC ec_mul_kernel(int B, const uint32_t* __restrict__ k, int kw, const uint32_t* __restrict__ P,
                              uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 s = ec::sc_reduce(k + (size_t)i * kw, kw);
  const ec::Jac r = P ? ec::jac_mul(s, ec::aff_load(P + (size_t)i * 16)) : ec::jac_mul_gen(s);
  ec::aff_store(out + (size_t)i * 16, ec::jac_to_aff(r));
}
__global__ void __launch_bounds__(64) MPE_EC_OCC ec_add_kernel(int B, const uint32_t* __restrict__ P, const uint32_t* __restrict__ Q,
                              uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::Jac r = ec::jac_add(ec::jac_from_aff(ec::aff_load(P + (size_t)i * 16)),
                                ec::jac_from_aff(ec::aff_load(Q + (size_t)i * 16)));
  ec::aff_store(out + (size_t)i * 16, ec::jac_to_aff(r));
}

// curv DLogProof: pk = sk G, R = rho G, c = H(R, G, pk) mod q, z = rho - c sk
// (point form and order of the three points: ec::Enc — recalled conventions are run-time properties of the context)
__device__ inline ec::U256 dlog_challenge(const ec::Aff& R, const ec::Aff& pk, const ec::Enc& enc) {
  ec::Sha256 s;
  ec::sha_init(s);
  const ec::Aff G = ec::aff_gen();
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int o = enc.ord_dlog[i];
    ec::sha_chain_point(s, o == 0 ? R : (o == 1 ? G : pk), enc);
  }
  const ec::U256 d = ec::sha_final(s);
  return ec::sc_reduce(d.w, 8);
}
__global__ void __launch_bounds__(64) MPE_EC_OCC dlog_prove_kernel(int B, ec::Enc enc, const uint32_t* __restrict__ sk, const uint32_t* __restrict__ nonce,
                                  uint32_t* __restrict__ pk, uint32_t* __restrict__ R, uint32_t* __restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 s = ec::sc_reduce(sk + (size_t)i * 8, 8), k = ec::sc_reduce(nonce + (size_t)i * 8, 8);
  const ec::Aff Rp = ec::jac_to_aff(ec::jac_mul_gen(k)), P = ec::jac_to_aff(ec::jac_mul_gen(s));
  const ec::U256 c = dlog_challenge(Rp, P, enc);
  ec::aff_store(pk + (size_t)i * 16, P);
  ec::aff_store(R + (size_t)i * 16, Rp);
  ec::u256_store(z + (size_t)i * 8, ec::sc_sub(k, ec::sc_mul(c, s)));
}
__global__ void __launch_bounds__(64) MPE_EC_OCC dlog_verify_kernel(int B, ec::Enc enc, const uint32_t* __restrict__ pk, const uint32_t* __restrict__ R,
                                   const uint32_t* __restrict__ z, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::Aff P = ec::aff_load(pk + (size_t)i * 16), Rp = ec::aff_load(R + (size_t)i * 16);
  if (!ec::aff_valid(P) || !ec::aff_valid(Rp)) { ok[i] = 0; return; }         // curv rejects such points when it deserialises them
  const ec::U256 c = dlog_challenge(Rp, P, enc), zz = ec::sc_reduce(z + (size_t)i * 8, 8);
  const ec::Jac l = ec::jac_add(ec::jac_mul_gen(zz), ec::jac_mul(c, P));
  ok[i] = ec::jac_eq_aff(l, Rp) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// glue kernels
// ---------------------------------------------------------------------------------------------
// r[nr] = a[na] * b[nb] + c[nc]      (plain integers; nr >= na + nb)
__global__ void muladd_kernel(int B, Rows a, int na, Rows b, int nb, Rows c, int nc, uint32_t* __restrict__ r, int nr) {
  MPE_FOREGROUND();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t x[90], y[90], t[180];
  sm::copy(x, row_of(a, i), na);
  sm::copy(y, row_of(b, i), nb);
  sm::mul(t, x, na, y, nb);
  for (int j = na + nb; j < nr; ++j) t[j] = 0;
  const uint32_t one[1] = {1};
  const uint32_t* cp = c.p ? row_of(c, i) : one;     // c.p == nullptr means the constant 1
  sm::add(t, nr, t, nr, cp, c.p ? nc : 1);           // callers size nr so that the sum fits
  uint32_t* o = r + (size_t)i * nr;
  for (int j = 0; j < nr; ++j) o[j] = t[j];
}

// Fiat-Shamir transcripts: SHA-256 over a list of fields, each hashed the way the reference hashes it
enum { HF_BIGINT = 0, HF_BIGINT_PLUS1 = 1, HF_POINT_COMPRESSED = 2 };
struct HashField { Rows r; int words; int kind; };
struct HashDesc { HashField f[14]; int n; };
__global__ void __launch_bounds__(64) MPE_EC_OCC hash_kernel(int B, ec::Enc enc, HashDesc d, uint32_t* __restrict__ out) {
  MPE_FOREGROUND();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  ec::Sha256 s;
  ec::sha_init(s);
  for (int k = 0; k < d.n; ++k) {
    const uint32_t* p = row_of(d.f[k].r, i);
    if (d.f[k].kind == HF_POINT_COMPRESSED) {
      ec::sha_point_compressed(s, ec::aff_load(p));
    } else if (d.f[k].kind == HF_BIGINT_PLUS1) {
      uint32_t t[130];
      const uint32_t one[1] = {1};
      t[d.f[k].words] = sm::add(t, d.f[k].words, p, d.f[k].words, one, 1);
      ec::sha_bigint(s, t, d.f[k].words + 1, enc);
    } else {
      ec::sha_bigint(s, p, d.f[k].words, enc);
    }
  }
  ec::u256_store(out + (size_t)i * 8, ec::sha_final(s));
}
static HashField hf(Rows r, int words, int kind = HF_BIGINT) { return HashField{r, words, kind}; }

// ok[i] = (a[i] <= q^3) for the s1 range check (range_proofs.rs:118 / :335)
__global__ void s1_range_kernel(int B, Rows s1, int words, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t q[8], q2[16], q3[24];
  for (int j = 0; j < 8; ++j) q[j] = ec::FQ[j];
  sm::mul(q2, q, 8, q, 8);
  sm::mul(q3, q2, 16, q, 8);
  ok[i] = sm::cmp(row_of(s1, i), words, q3, 24) <= 0 ? 1 : 0;
}
// ok[i] &= all of: flags a, b (optional) and equality of the [words] rows x == y (optional)
__global__ void and_flags_kernel(int B, uint8_t* __restrict__ ok, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                 const uint32_t* __restrict__ x, Rows y, int words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  bool v = ok[i] != 0;
  if (a) v = v && a[i];
  if (b) v = v && b[i];
  if (x) v = v && sm::cmp(x + (size_t)i * words, words, row_of(y, i), words) == 0;
  ok[i] = v ? 1 : 0;
}
__global__ void fill_u8_kernel(int B, uint8_t* p, uint8_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) p[i] = v;
}
// PDL verify: ok &= ( (s1 mod q) G + (q - e) Q == u1 )      (zk_pdl_with_slack/mod.rs:138-142,174)
__global__ void __launch_bounds__(64) MPE_EC_OCC pdl_u1_check_kernel(int B, Rows s1, const uint32_t* __restrict__ e, Rows G, Rows Q, Rows u1,
                                    uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 a = ec::sc_reduce(row_of(s1, i), 25);
  const ec::U256 ne = ec::sc_neg(ec::sc_reduce(e + (size_t)i * 8, 8));
  const ec::Aff Gp = ec::aff_load(row_of(G, i)), Qp = ec::aff_load(row_of(Q, i)), U1 = ec::aff_load(row_of(u1, i));
  if (!ec::aff_valid(Gp) || !ec::aff_valid(Qp) || !ec::aff_valid(U1)) { ok[i] = 0; return; }   // statement / proof points come from a peer
  const ec::Jac l = ec::jac_add(ec::jac_mul(a, Gp), ec::jac_mul(ne, Qp));
  if (!ec::jac_eq_aff(l, U1)) ok[i] = 0;
}
// out = (k mod q) * P with per-item rows (P.p == nullptr -> generator)
__global__ void __launch_bounds__(64) MPE_EC_OCC ec_mul_rows_kernel(int B, Rows k, int kw, Rows P, uint32_t* __restrict__ out) {
  MPE_FOREGROUND();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 s = ec::sc_reduce(row_of(k, i), kw);
  const ec::Jac r = P.p ? ec::jac_mul(s, ec::aff_load(row_of(P, i))) : ec::jac_mul_gen(s);
  ec::aff_store(out + (size_t)i * 16, ec::jac_to_aff(r));
}

// ---------------------------------------------------------------------------------------------
// host-side sequencing helper: every intermediate lives in the context workspace
// ---------------------------------------------------------------------------------------------
struct Seq {
  mpe_ctx* ctx;
  hipStream_t st;
  int B;
  int rc = MPE_OK;
  uint32_t* words(size_t per_item) {
    uint32_t* p = ws_array<uint32_t>(ctx, (size_t)B * per_item);
    if (!p && rc == MPE_OK) { rc = MPE_E_NOMEM; mpe_set_error_msg("workspace under-reserved"); }
    return p;
  }
  uint8_t* flags() {
    uint8_t* p = ws_array<uint8_t>(ctx, (size_t)B);
    if (!p && rc == MPE_OK) { rc = MPE_E_NOMEM; mpe_set_error_msg("workspace under-reserved"); }
    return p;
  }
  // h1^x / h2^x mod N~ of a statement: fixed-base tables when the statement set has them (mpe_fixedbase.h)
  uint32_t* fb_modexp(const mpe_statements* stm, Rows sel, int which, Rows base, Rows exps, int ew) {
    if (!stm->fb_tab || !ctx->use_fixed_base) return modexp(stm->ms, sel, base, exps, ew);
    uint32_t* o = words(64);
    if (rc == MPE_OK) rc = launch_fb_modexp(ctx, stm, B, sel, which, exps, ew, o, st);
    return o;
  }
  uint32_t* modexp(const mpe_modset* ms, Rows sel, Rows base, Rows exps, int ew) {
    uint32_t* o = words(ms->bits / 32);
    if (rc == MPE_OK) rc = launch_modexp(ctx, ms, B, sel, base, no_rows(), exps, ew, o, st);
    return o;
  }
  // x^e mod N^2; holder = this party owns the key and may go through p^2 | q^2
  uint32_t* modexp_nn(const mpe_paillier* pk, Rows sel, Rows base, Rows exps, int ew, bool holder, bool pow_n = false) {
    uint32_t* o = words(128);
    if (rc == MPE_OK) rc = mpe::modexp_nn(ctx, pk, B, sel, base, exps, ew, holder, o, st, pow_n);
    return o;
  }
  uint32_t* modexp_nn2(const mpe_paillier* pk, Rows sel, Rows base, Rows exps, int ew, Rows base2, Rows exps2, int ew2) {
    uint32_t* o = words(128);
    if (rc == MPE_OK) rc = mpe::modexp_nn2(ctx, pk, B, sel, base, exps, ew, base2, exps2, ew2, o, st);
    return o;
  }
  // x^e mod N of the key's holder: through p | q when the context may (modexp_n_holder), else the 2048-bit ladder
  uint32_t* modexp_n(const mpe_paillier* pk, const int32_t* key_idx, Rows sel, Rows base, Rows exps, int ew) {
    if (!(pk->has_private && ctx->use_crt && ctx->use_pair && ctx->use_crt_n)) return modexp(pk->ms_n, sel, base, exps, ew);
    uint32_t* o = words(64);
    if (rc == MPE_OK) rc = mpe::modexp_n_holder(ctx, pk, B, key_idx, base, exps, ew, o, st);
    return o;
  }
  uint32_t* modmul(const mpe_modset* ms, Rows sel, Rows a, Rows b) {
    uint32_t* o = words(ms->bits / 32);
    if (rc == MPE_OK) rc = launch_modmul(ctx, ms, B, sel, a, b, o, st);
    return o;
  }
  void modmul_to(const mpe_modset* ms, Rows sel, Rows a, Rows b, uint32_t* o) {
    if (rc == MPE_OK) rc = launch_modmul(ctx, ms, B, sel, a, b, o, st);
  }
  uint32_t* modinv(const mpe_modset* ms, Rows sel, Rows a, uint8_t* ok) {
    uint32_t* o = words(ms->bits / 32);
    if (rc == MPE_OK) rc = launch_modinv(ctx, ms, B, sel, a, o, ok, st);
    return o;
  }
  void muladd(Rows a, int na, Rows b, int nb, Rows c, int nc, uint32_t* r, int nr) {
    if (rc != MPE_OK) return;
    hipLaunchKernelGGL(muladd_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, a, na, b, nb, c, nc, r, nr);
  }
  void hash(const HashDesc& d, uint32_t* out) {
    if (rc != MPE_OK) return;
    hipLaunchKernelGGL(hash_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, ctx->enc, d, out);
  }
  int finish(const char* what) {
    if (rc != MPE_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mpe_set_error(what, e); return MPE_E_HIP; }
    return MPE_OK;
  }
};

// per-item selectors of per-key / per-statement tables
static Rows sel_of(const int32_t* idx, int count) { return Rows{nullptr, idx, (idx == nullptr && count != 1) ? 1 : 0, 0}; }
static Rows tab_rows(const uint32_t* table, int stride, const int32_t* idx, int count, int words = 0) {
  if (idx) return Rows{table, idx, stride, words};
  return Rows{table, nullptr, count == 1 ? 0 : stride, words};
}

// proofs read through Rows so that the round pipeline can verify a proof in place, without gathering
struct AliceProofRows { Rows z, e, s, s1, s2; };
struct PdlProofRows { Rows z, u1, u2, u3, s1, s2, s3; };
static AliceProofRows dense(const mpe_alice_proof* p) {
  return AliceProofRows{rows(p->z, 64), rows(p->e, 8), rows(p->s, 64), rows(p->s1, 25), rows(p->s2, 89)};
}
static PdlProofRows dense(const mpe_pdl_proof* p) {
  return PdlProofRows{rows(p->z, 64), rows(p->u1, 16), rows(p->u2, 128), rows(p->u3, 64), rows(p->s1, 25), rows(p->s2, 64),
                      rows(p->s3, 89)};
}
static Rows with_words(Rows r, int words) { r.words = words; return r; }

// ---------------------------------------------------------------------------------------------
// AliceProof::generate   (range_proofs.rs:160-193; rounds :39-67 and :78-90)
// Small batches: z, u and w are independent and run on three streams (mpe::Fork); `outer`: a fork of the CALLER that
// produces `cipher` concurrently (Round 0 encrypts k_i on another stream) — joined right before the transcript hash.
// ---------------------------------------------------------------------------------------------
static inline size_t ws_need_alice_generate(int B) { return (size_t)B * (1400 + CRT_WS_WORDS + MODEXP_N_HOLDER_WS_WORDS) * 4 + 65536; }
static inline size_t ws_need_alice_verify(int B) { return (size_t)B * 2200 * 4 + 65536; }
static int merge_rc(const Seq& a, const Seq& b, const Seq& c) { return a.rc != MPE_OK ? a.rc : (b.rc != MPE_OK ? b.rc : c.rc); }

static int alice_generate(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int B, const int32_t* key_idx,
                          const int32_t* st_idx, Rows a, Rows cipher, Rows r,
                          const mpe_alice_nonces* nn, const mpe_alice_proof* out, hipStream_t st, Fork* outer = nullptr,
                          const uint32_t* bn_pre = nullptr, hipEvent_t bn_pre_ready = nullptr,        // bn_pre: beta^N mod N^2 when the caller
                          hipEvent_t cipher_ready = nullptr) {                                        // already has it (or has queued it
                                                                                                      // elsewhere: bn_pre_ready);
                                                                                                      // cipher_ready: the outer branch goes on
                                                                                                      // with other work — wait for this event
                                                                                                      // instead of joining it
  MPE_TRY(ws_reserve(ctx, ws_need_alice_generate(B), st));
  Fork f(ctx, st, 3, B <= ctx->par_items);
  Seq q{ctx, f.s(0), B}, q1{ctx, f.s(1), B}, q2{ctx, f.s(2), B};
  const Rows ksel = sel_of(key_idx, pk->nkeys), ssel = sel_of(st_idx, stm->count);
  const Rows h1 = tab_rows(stm->h1, 64, st_idx, stm->count), h2 = tab_rows(stm->h2, 64, st_idx, stm->count);
  const Rows Nrow = tab_rows(pk->N, 64, key_idx, pk->nkeys);
  // z = h1^a h2^rho mod N~                                                      :52
  uint32_t* z1 = q1.fb_modexp(stm, ssel, 0, h1,a, 8);
  uint32_t* z2 = q1.fb_modexp(stm, ssel, 1, h2,rows(nn->rho, 72), 72);
  q1.modmul_to(stm->ms, ssel, rows(z1, 64), rows(z2, 64), out->z);
  // u = (alpha N + 1) beta^N mod N^2                                            :53-55
  uint32_t* gu = q.words(128);
  q.muladd(rows(nn->alpha, 24), 24, Nrow, 64, no_rows(), 0, gu, 128);
  if (bn_pre && bn_pre_ready) (void)hipStreamWaitEvent(f.s(0), bn_pre_ready, 0);
  const uint32_t* bn = bn_pre ? bn_pre : q.modexp_nn(pk, ksel, rows(nn->beta, 64, nullptr, 64), Nrow, 64, true, true);   // the prover owns the key
  uint32_t* u = q.modmul(pk->ms_nn, ksel, rows(gu, 128), rows(bn, 128));
  // w = h1^alpha h2^gamma mod N~                                                :56-57
  uint32_t* w1 = q2.fb_modexp(stm, ssel, 0, h1,rows(nn->alpha, 24), 24);
  uint32_t* w2 = q2.fb_modexp(stm, ssel, 1, h2,rows(nn->gamma, 88), 88);
  uint32_t* w = q2.modmul(stm->ms, ssel, rows(w1, 64), rows(w2, 64));
  f.join();
  if (cipher_ready) (void)hipStreamWaitEvent(st, cipher_ready, 0);
  else if (outer) outer->join();
  q.rc = merge_rc(q, q1, q2);
  // e = H(N, N+1, c, z, u, w)                                                   :175-182
  HashDesc d;
  d.n = 6;
  d.f[0] = hf(Nrow, 64); d.f[1] = hf(Nrow, 64, HF_BIGINT_PLUS1); d.f[2] = hf(cipher, 128);
  d.f[3] = hf(rows(out->z, 64), 64); d.f[4] = hf(rows(u, 128), 128); d.f[5] = hf(rows(w, 64), 64);
  q.hash(d, out->e);
  // s = r^e beta mod N ; s1 = e a + alpha ; s2 = e rho + gamma                  :84-88
  uint32_t* re = q.modexp_n(pk, key_idx, ksel, r, rows(out->e, 8), 8);
  q.modmul_to(pk->ms_n, ksel, rows(re, 64), rows(nn->beta, 64), out->s);
  q.muladd(rows(out->e, 8), 8, a, 8, rows(nn->alpha, 24), 24, out->s1, 25);
  q.muladd(rows(out->e, 8), 8, rows(nn->rho, 72), 72, rows(nn->gamma, 88), 88, out->s2, 89);
  return q.finish("alice_generate");
}

// ---------------------------------------------------------------------------------------------
// AliceProof::verify   (range_proofs.rs:105-156).  Small batches: the N~ side and the N^2 side on separate streams.
// ---------------------------------------------------------------------------------------------
// m_pre / inv_ok_pre: the caller already holds m = s^N (c^-1)^e mod N^2 and the verdict of the inversion of c (Round 1 computes the
// ladders of its verifications and of its MessageBs in ONE launch, mpe_gg20.h round1_merged_ladders)
static int alice_verify(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int B, const int32_t* key_idx,
                        const int32_t* st_idx, Rows cipher, const AliceProofRows& pr, uint8_t* ok, hipStream_t st,
                        const uint32_t* m_pre = nullptr, const uint8_t* inv_ok_pre = nullptr, hipEvent_t m_pre_ready = nullptr) {
  // m_pre: s^N (c^-1)^e already computed (or queued on another stream: then m_pre_ready says when) by the caller's merged ladder launch
  MPE_TRY(ws_reserve(ctx, ws_need_alice_verify(B), st));
  Fork f(ctx, st, 3, B <= ctx->par_items);          // (with m_pre the N^2 side is one multiplication: the two N~ branches still fork)
  Seq q{ctx, f.s(0), B}, q1{ctx, f.s(1), B}, q2{ctx, f.s(2), B};
  const Rows ksel = sel_of(key_idx, pk->nkeys), ssel = sel_of(st_idx, stm->count);
  const Rows h1 = tab_rows(stm->h1, 64, st_idx, stm->count), h2 = tab_rows(stm->h2, 64, st_idx, stm->count);
  const Rows Nrow = tab_rows(pk->N, 64, key_idx, pk->nkeys);
  MPE_LAUNCH_1D(s1_range_kernel, B, st, B, pr.s1, 25, ok);                                         // :118
  // w' = h1^s1 h2^s2 (z^e)^-1 mod N~                                                               :122-132
  uint8_t *inv_ok1 = q.flags(), *inv_ok2 = q.flags();
  uint32_t* ze = q1.modexp(stm->ms, ssel, pr.z, pr.e, 8);
  uint32_t* zei = q1.modinv(stm->ms, ssel, rows(ze, 64), inv_ok1);
  uint32_t* a1 = q2.fb_modexp(stm, ssel, 0, h1,pr.s1, 25);
  uint32_t* a2 = q2.fb_modexp(stm, ssel, 1, h2,pr.s2, 89);
  uint32_t* a12 = q2.modmul(stm->ms, ssel, rows(a1, 64), rows(a2, 64));
  // u' = (s1 N + 1) s^N (c^e)^-1 mod N^2                                                           :134-141
  uint32_t* gs1 = q.words(128);
  q.muladd(pr.s1, 25, Nrow, 64, no_rows(), 0, gs1, 128);
  uint32_t *u = nullptr, *b12 = nullptr, *cie = nullptr;
  if (m_pre) {
    inv_ok2 = const_cast<uint8_t*>(inv_ok_pre);
    if (m_pre_ready) (void)hipStreamWaitEvent(f.s(0), m_pre_ready, 0);
    u = q.modmul(pk->ms_nn, ksel, rows(gs1, 128), rows(m_pre, 128));
  } else if (f.on) {
    // small batch (latency-bound): the 2048-bit ladder s^N starts at once; the inversion of c and the short ladder
    // (c^-1)^e run beside it on the stream of the fixed-base side, and one more multiplication joins them
    uint32_t* sn = q.modexp_nn(pk, ksel, with_words(pr.s, 64), Nrow, 64, false);
    b12 = q.modmul(pk->ms_nn, ksel, rows(gs1, 128), rows(sn, 128));
    uint32_t* cred = q2.modmul(pk->ms_nn, ksel, cipher, rows(pk->ms_nn->one_words, 0, nullptr, 1));   // c mod N^2
    uint32_t* cinv = q2.modinv(pk->ms_nn, ksel, rows(cred, 128), inv_ok2);
    cie = q2.modexp_nn(pk, ksel, rows(cinv, 128), pr.e, 8, false);
  } else if (ctx->use_multiexp) {
    // (c^e)^-1 = (c^-1)^e: invert first, then s^N (c^-1)^e on one ladder (the 256 squarings of c^e are shared)
    uint32_t* cred = q.modmul(pk->ms_nn, ksel, cipher, rows(pk->ms_nn->one_words, 0, nullptr, 1));   // c mod N^2
    uint32_t* cinv = q.modinv(pk->ms_nn, ksel, rows(cred, 128), inv_ok2);
    uint32_t* m = q.modexp_nn2(pk, ksel, with_words(pr.s, 64), Nrow, 64, rows(cinv, 128), pr.e, 8);
    u = q.modmul(pk->ms_nn, ksel, rows(gs1, 128), rows(m, 128));
  } else {
    uint32_t* ce = q.modexp_nn(pk, ksel, cipher, pr.e, 8, false);
    uint32_t* cei = q.modinv(pk->ms_nn, ksel, rows(ce, 128), inv_ok2);
    uint32_t* sn = q.modexp_nn(pk, ksel, with_words(pr.s, 64), Nrow, 64, false);
    uint32_t* b12n = q.modmul(pk->ms_nn, ksel, rows(gs1, 128), rows(sn, 128));
    u = q.modmul(pk->ms_nn, ksel, rows(b12n, 128), rows(cei, 128));
  }
  f.join();
  q.rc = merge_rc(q, q1, q2);
  if (f.on && !m_pre) u = q.modmul(pk->ms_nn, ksel, rows(b12, 128), rows(cie, 128));
  uint32_t* w = q.modmul(stm->ms, ssel, rows(a12, 64), rows(zei, 64));
  // e' = H(N, N+1, c, z, u', w') == e                                                              :143-153
  uint32_t* e2 = q.words(8);
  HashDesc d;
  d.n = 6;
  d.f[0] = hf(Nrow, 64); d.f[1] = hf(Nrow, 64, HF_BIGINT_PLUS1); d.f[2] = hf(cipher, 128);
  d.f[3] = hf(pr.z, 64); d.f[4] = hf(rows(u, 128), 128); d.f[5] = hf(rows(w, 64), 64);
  q.hash(d, e2);
  if (q.rc == MPE_OK)
    hipLaunchKernelGGL(and_flags_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, ok, inv_ok1, inv_ok2, e2, pr.e, 8);
  return q.finish("alice_verify");
}

// ---------------------------------------------------------------------------------------------
// PDLwSlackProof::prove   (zk_pdl_with_slack/mod.rs:68-125)
// ---------------------------------------------------------------------------------------------
static int pdl_prove(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int B, const int32_t* key_idx,
                     const int32_t* st_idx, Rows cipher, Rows Qp, Rows Gp, Rows x, Rows r, const mpe_pdl_nonces* nn,
                     const mpe_pdl_proof* out, hipStream_t st, Fork* outer = nullptr, const uint32_t* bn_pre = nullptr,
                     hipEvent_t bn_pre_ready = nullptr, hipEvent_t G_ready = nullptr) {      // G_ready: G is there although branch 1 goes on
  // outer: a fork of the CALLER whose branch 1 produces the statement points Q, G concurrently (Round 4: R, R_dash); only u1
  // and the transcript hash need them.  bn_pre: beta^N mod N^2 when the caller has queued it elsewhere (done at bn_pre_ready)
  MPE_TRY(ws_reserve(ctx, (size_t)B * (1400 + CRT_WS_WORDS + MODEXP_N_HOLDER_WS_WORDS) * 4 + 65536, st));
  Fork f(ctx, st, 3, B <= ctx->par_items);
  Seq q{ctx, f.s(0), B}, q1{ctx, f.s(1), B}, q2{ctx, f.s(2), B};
  const Rows ksel = sel_of(key_idx, pk->nkeys), ssel = sel_of(st_idx, stm->count);
  const Rows h1 = tab_rows(stm->h1, 64, st_idx, stm->count), h2 = tab_rows(stm->h2, 64, st_idx, stm->count);
  const Rows Nrow = tab_rows(pk->N, 64, key_idx, pk->nkeys);
  // z = h1^x h2^rho mod N~                                                      :79-85
  uint32_t* z1 = q1.fb_modexp(stm, ssel, 0, h1,x, 8);
  uint32_t* z2 = q1.fb_modexp(stm, ssel, 1, h2,rows(nn->rho, 72), 72);
  q1.modmul_to(stm->ms, ssel, rows(z1, 64), rows(z2, 64), out->z);
  // u1 = (alpha mod q) G                                                        :86
  if (G_ready) (void)hipStreamWaitEvent(f.s(1), G_ready, 0);
  else if (outer) outer->branch_done_wait(1, f.s(1));
  if (B > 0) hipLaunchKernelGGL(ec_mul_rows_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, f.s(1), B, rows(nn->alpha, 24), 24, Gp, out->u1);
  // u2 = (N+1)^alpha beta^N mod N^2; (N+1)^alpha = 1 + alpha N (mod N^2), alpha N + 1 < N^2      :87-93
  uint32_t* ga = q.words(128);
  q.muladd(rows(nn->alpha, 24), 24, Nrow, 64, no_rows(), 0, ga, 128);
  if (bn_pre && bn_pre_ready) (void)hipStreamWaitEvent(f.s(0), bn_pre_ready, 0);
  const uint32_t* bn = bn_pre ? bn_pre : q.modexp_nn(pk, ksel, rows(nn->beta, 64, nullptr, 64), Nrow, 64, true, true);   // the prover owns the key
  q.modmul_to(pk->ms_nn, ksel, rows(ga, 128), rows(bn, 128), out->u2);
  // u3 = h1^alpha h2^gamma mod N~                                               :94-100
  uint32_t* w1 = q2.fb_modexp(stm, ssel, 0, h1,rows(nn->alpha, 24), 24);
  uint32_t* w2 = q2.fb_modexp(stm, ssel, 1, h2,rows(nn->gamma, 88), 88);
  q2.modmul_to(stm->ms, ssel, rows(w1, 64), rows(w2, 64), out->u3);
  f.join();
  if (outer) outer->join();
  q.rc = merge_rc(q, q1, q2);
  // e = H(G, Q, c, z, u1, u2, u3)                                               :102-110
  uint32_t* e = q.words(8);
  HashDesc d;
  d.n = 7;
  d.f[0] = hf(Gp, 16, HF_POINT_COMPRESSED); d.f[1] = hf(Qp, 16, HF_POINT_COMPRESSED);
  d.f[2] = hf(cipher, 128); d.f[3] = hf(rows(out->z, 64), 64);
  d.f[4] = hf(rows(out->u1, 16), 16, HF_POINT_COMPRESSED); d.f[5] = hf(rows(out->u2, 128), 128);
  d.f[6] = hf(rows(out->u3, 64), 64);
  q.hash(d, e);
  // s1 = e x + alpha ; s2 = r^e beta mod N ; s3 = e rho + gamma                :112-114
  q.muladd(rows(e, 8), 8, x, 8, rows(nn->alpha, 24), 24, out->s1, 25);
  uint32_t* re = q.modexp_n(pk, key_idx, ksel, r, rows(e, 8), 8);
  q.modmul_to(pk->ms_n, ksel, rows(re, 64), rows(nn->beta, 64), out->s2);
  q.muladd(rows(e, 8), 8, rows(nn->rho, 72), 72, rows(nn->gamma, 88), 88, out->s3, 89);
  return q.finish("pdl_prove");
}

// ---------------------------------------------------------------------------------------------
// PDLwSlackProof::verify   (zk_pdl_with_slack/mod.rs:127-179).  Small batches: the EC check, the N^2 side and the N~ side
// run on three streams once the challenge is known.
// ---------------------------------------------------------------------------------------------
static int pdl_verify(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int B, const int32_t* key_idx,
                      const int32_t* st_idx, Rows cipher, Rows Qp, Rows Gp, const PdlProofRows& pr, uint8_t* ok,
                      hipStream_t st) {
  MPE_TRY(ws_reserve(ctx, (size_t)B * 2600 * 4 + 65536, st));
  const Rows ksel = sel_of(key_idx, pk->nkeys), ssel = sel_of(st_idx, stm->count);
  const Rows h1 = tab_rows(stm->h1, 64, st_idx, stm->count), h2 = tab_rows(stm->h2, 64, st_idx, stm->count);
  const Rows Nrow = tab_rows(pk->N, 64, key_idx, pk->nkeys);
  Seq q0{ctx, st, B};
  uint32_t* e = q0.words(8);
  HashDesc d;
  d.n = 7;
  d.f[0] = hf(Gp, 16, HF_POINT_COMPRESSED); d.f[1] = hf(Qp, 16, HF_POINT_COMPRESSED);
  d.f[2] = hf(cipher, 128); d.f[3] = hf(pr.z, 64);
  d.f[4] = hf(pr.u1, 16, HF_POINT_COMPRESSED); d.f[5] = hf(pr.u2, 128);
  d.f[6] = hf(pr.u3, 64);
  q0.hash(d, e);                                                                                    // :128-136
  Fork f(ctx, st, 3, B <= ctx->par_items);
  Seq q{ctx, f.s(0), B}, q1{ctx, f.s(1), B}, q2{ctx, f.s(2), B};
  q.rc = q0.rc;
  if (B > 0) {
    hipLaunchKernelGGL(fill_u8_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, f.s(1), B, ok, (uint8_t)1);
    hipLaunchKernelGGL(pdl_u1_check_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, f.s(1), B, pr.s1, e, Gp, Qp, pr.u1, ok);   // :138-142
  }
  uint8_t *inv_ok1 = q.flags(), *inv_ok2 = q.flags();
  // u2' = (N+1)^s1 s2^N c^-e mod N^2; (N+1)^s1 = 1 + s1 N < N^2 because s1 < 2^800                 :144-157
  uint32_t* g1 = q.words(128);
  q.muladd(pr.s1, 25, Nrow, 64, no_rows(), 0, g1, 128);
  uint32_t *u2 = nullptr, *t2s = nullptr, *cies = nullptr;
  if (f.on) {
    // small batch: s2^N starts at once; c^-1 and the short ladder (c^-1)^e follow the u1 check on its stream
    uint32_t* s2n = q.modexp_nn(pk, ksel, with_words(pr.s2, 64), Nrow, 64, false);
    t2s = q.modmul(pk->ms_nn, ksel, rows(g1, 128), rows(s2n, 128));
    uint32_t* cred = q1.modmul(pk->ms_nn, ksel, cipher, rows(pk->ms_nn->one_words, 0, nullptr, 1));  // c mod N^2
    uint32_t* cinv = q1.modinv(pk->ms_nn, ksel, rows(cred, 128), inv_ok1);
    cies = q1.modexp_nn(pk, ksel, rows(cinv, 128), rows(e, 8), 8, false);
  } else {
    uint32_t* cred = q.modmul(pk->ms_nn, ksel, cipher, rows(pk->ms_nn->one_words, 0, nullptr, 1));  // c mod N^2
    uint32_t* cinv = q.modinv(pk->ms_nn, ksel, rows(cred, 128), inv_ok1);
    if (ctx->use_multiexp) {
      uint32_t* m = q.modexp_nn2(pk, ksel, with_words(pr.s2, 64), Nrow, 64, rows(cinv, 128), rows(e, 8), 8);
      u2 = q.modmul(pk->ms_nn, ksel, rows(g1, 128), rows(m, 128));
    } else {
      uint32_t* s2n = q.modexp_nn(pk, ksel, with_words(pr.s2, 64), Nrow, 64, false);
      uint32_t* t2 = q.modmul(pk->ms_nn, ksel, rows(g1, 128), rows(s2n, 128));
      uint32_t* cie = q.modexp_nn(pk, ksel, rows(cinv, 128), rows(e, 8), 8, false);
      u2 = q.modmul(pk->ms_nn, ksel, rows(t2, 128), rows(cie, 128));
    }
  }
  // u3' = h1^s1 h2^s3 z^-e mod N~                                                                  :159-172
  uint32_t* a1 = q2.fb_modexp(stm, ssel, 0, h1,pr.s1, 25);
  uint32_t* a2 = q2.fb_modexp(stm, ssel, 1, h2,pr.s3, 89);
  uint32_t* a12 = q2.modmul(stm->ms, ssel, rows(a1, 64), rows(a2, 64));
  uint32_t* zred = q2.modmul(stm->ms, ssel, pr.z, rows(stm->ms->one_words, 0, nullptr, 1));
  uint32_t* zinv = q2.modinv(stm->ms, ssel, rows(zred, 64), inv_ok2);
  uint32_t* zie = q2.modexp(stm->ms, ssel, rows(zinv, 64), rows(e, 8), 8);
  uint32_t* u3 = q2.modmul(stm->ms, ssel, rows(a12, 64), rows(zie, 64));
  f.join();
  q.rc = merge_rc(q, q1, q2);
  if (f.on) u2 = q.modmul(pk->ms_nn, ksel, rows(t2s, 128), rows(cies, 128));
  if (q.rc == MPE_OK) {                                                                             // :174
    hipLaunchKernelGGL(and_flags_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, ok, inv_ok1, inv_ok2, u2, pr.u2, 128);
    hipLaunchKernelGGL(and_flags_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, ok, (const uint8_t*)nullptr,
                       (const uint8_t*)nullptr, u3, pr.u3, 64);
  }
  return q.finish("pdl_verify");
}

}  // namespace mpe

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int mpe_statements_create(mpe_ctx* ctx, int count, const uint32_t* d_Nt, const uint32_t* d_h1, const uint32_t* d_h2,
                          mpe_statements** out, void* stream) {
  if (!ctx) return MPE_E_ARG;
  return mpe_statements_create_wb(ctx, count, d_Nt, d_h1, d_h2, ctx->fb_window_bits, out, stream);
}
int mpe_statements_create_wb(mpe_ctx* ctx, int count, const uint32_t* d_Nt, const uint32_t* d_h1, const uint32_t* d_h2, int wb,
                             mpe_statements** out, void* stream) {
  if (!ctx || !d_Nt || !d_h1 || !d_h2 || !out || count <= 0 || (wb != 0 && wb < 2) || wb > 16) return MPE_E_ARG;   // wb == 0: one-off statements, no tables
  hipStream_t st = (hipStream_t)stream;
  mpe_statements* s = new (std::nothrow) mpe_statements();
  if (!s) return MPE_E_NOMEM;
  s->count = count;
  const size_t w = (size_t)count * 64;
  hipError_t e = hipMalloc(&s->blob, 3 * w * 4);
  if (e != hipSuccess) { delete s; mpe_set_error("hipMalloc(statements)", e); return MPE_E_NOMEM; }
  s->Nt = (uint32_t*)s->blob; s->h1 = s->Nt + w; s->h2 = s->h1 + w;
  (void)hipMemcpyAsync(s->Nt, d_Nt, w * 4, hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(s->h1, d_h1, w * 4, hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(s->h2, d_h2, w * 4, hipMemcpyDeviceToDevice, st);
  int rc = mpe::modset_create_dev(ctx, 2048, count, s->Nt, &s->ms, st);
  if (rc != MPE_OK) { (void)hipFree(s->blob); delete s; return rc; }
  s->fb_wb = 0;                                  // no tables unless they are built below
  if (ctx->use_fixed_base && wb != 0) {
    // fixed-base window tables of h1, h2 (26 MB per base at 8-bit windows), built on the GPU once per statement set:
    // the window bases one after the other (squarings), then every window's multiples in parallel
    using C = mpe::Cfg2048;
    s->fb_wb = wb;
    const int nwindows = mpe::fb_windows(s->fb_wb);
    const size_t bytes = (size_t)2 * count * nwindows * ((size_t)1 << s->fb_wb) * C::K * sizeof(uint32_t);
    e = hipMalloc((void**)&s->fb_tab, bytes);
    if (e != hipSuccess) { mpe_set_error("hipMalloc(fixed-base tables)", e); mpe_statements_destroy(s); return MPE_E_NOMEM; }
    mpe::ModsetView v;
    v.n_limbs = s->ms->n_limbs; v.one_limbs = s->ms->one_limbs; v.r2_limbs = s->ms->r2_limbs; v.r2h_limbs = s->ms->r2h_limbs;
    v.n0inv = s->ms->n0inv; v.count = s->ms->count;
    const int npairs = 2 * count, nrows = npairs * nwindows;
    hipLaunchKernelGGL(mpe::fb_bases_kernel<C>, dim3((npairs + C::GROUPS - 1) / C::GROUPS), dim3(64), 0, st, npairs, v, s->h1,
                       s->h2, s->fb_wb, s->fb_tab);
    hipLaunchKernelGGL(mpe::fb_fill_kernel<C>, dim3((nrows + C::GROUPS - 1) / C::GROUPS), dim3(64), 0, st, nrows, v, s->fb_wb,
                       s->fb_tab);
    e = hipGetLastError();
    if (e != hipSuccess) { mpe_set_error("fixed-base table build", e); mpe_statements_destroy(s); return MPE_E_HIP; }
  }
  *out = s;
  return MPE_OK;
}
int mpe_statements_destroy(mpe_statements* s) {
  if (!s) return MPE_E_ARG;
  if (s->ms) mpe_modset_destroy(s->ms);
  if (s->fb_tab) (void)hipFree(s->fb_tab);
  if (s->blob) (void)hipFree(s->blob);
  delete s;
  return MPE_OK;
}

int mpe_ec_mul_base(mpe_ctx* ctx, int batch, const uint32_t* d_k, int k_words, uint32_t* d_out, void* stream) {
  if (!ctx || !d_k || !d_out || batch < 0 || k_words <= 0 || k_words > 89) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::ec_mul_kernel, batch, st, batch, d_k, k_words, (const uint32_t*)nullptr, d_out);
  return MPE_OK;
}
int mpe_ec_mul(mpe_ctx* ctx, int batch, const uint32_t* d_k, int k_words, const uint32_t* d_P, uint32_t* d_out,
               void* stream) {
  if (!ctx || !d_k || !d_P || !d_out || batch < 0 || k_words <= 0 || k_words > 89) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::ec_mul_kernel, batch, st, batch, d_k, k_words, d_P, d_out);
  return MPE_OK;
}
int mpe_ec_add(mpe_ctx* ctx, int batch, const uint32_t* d_P, const uint32_t* d_Q, uint32_t* d_out, void* stream) {
  if (!ctx || !d_P || !d_Q || !d_out || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::ec_add_kernel, batch, st, batch, d_P, d_Q, d_out);
  return MPE_OK;
}
int mpe_dlog_prove(mpe_ctx* ctx, int batch, const uint32_t* d_sk, const uint32_t* d_nonce, uint32_t* d_pk,
                   uint32_t* d_R, uint32_t* d_z, void* stream) {
  if (!ctx || !d_sk || !d_nonce || !d_pk || !d_R || !d_z || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::dlog_prove_kernel, batch, st, batch, ctx->enc, d_sk, d_nonce, d_pk, d_R, d_z);
  return MPE_OK;
}
int mpe_dlog_verify(mpe_ctx* ctx, int batch, const uint32_t* d_pk, const uint32_t* d_R, const uint32_t* d_z,
                    uint8_t* d_ok, void* stream) {
  if (!ctx || !d_pk || !d_R || !d_z || !d_ok || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::dlog_verify_kernel, batch, st, batch, ctx->enc, d_pk, d_R, d_z, d_ok);
  return MPE_OK;
}
int mpe_modinv(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx, const uint32_t* d_a,
               uint32_t* d_out, uint8_t* d_ok, void* stream) {
  if (!ctx || !ms || !d_a || !d_out || !d_ok || batch < 0) return MPE_E_ARG;
  if (!d_mod_idx && ms->count != 1 && ms->count < batch) return MPE_E_ARG;
  MPE_TRY(mpe::ws_reserve(ctx, mpe::modinv_ws_words(ms, batch) * 4, (hipStream_t)stream));
  return mpe::launch_modinv(ctx, ms, batch, mpe::sel_of(d_mod_idx, ms->count), mpe::rows(d_a, ms->bits / 32), d_out, d_ok,
                            (hipStream_t)stream);
}

static bool proof_args_ok(const mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                          const int32_t* key_idx, const int32_t* st_idx) {
  if (!ctx || !pk || !stm || batch < 0) return false;
  if (!key_idx && pk->nkeys != 1 && pk->nkeys < batch) return false;
  if (!st_idx && stm->count != 1 && stm->count < batch) return false;
  return true;
}
int mpe_alice_generate(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch,
                       const int32_t* d_key_idx, const int32_t* d_st_idx, const uint32_t* d_a, const uint32_t* d_cipher,
                       const uint32_t* d_r, const mpe_alice_nonces* nonces, const mpe_alice_proof* out, void* stream) {
  if (!proof_args_ok(ctx, pk, stm, batch, d_key_idx, d_st_idx) || !d_a || !d_cipher || !d_r || !nonces || !out) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  return mpe::alice_generate(ctx, pk, stm, batch, d_key_idx, d_st_idx, mpe::rows(d_a, 8), mpe::rows(d_cipher, 128),
                             mpe::rows(d_r, 64), nonces, out, (hipStream_t)stream);
}
int mpe_alice_verify(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch, const int32_t* d_key_idx,
                     const int32_t* d_st_idx, const uint32_t* d_cipher, const mpe_alice_proof* proof, uint8_t* d_ok,
                     void* stream) {
  if (!proof_args_ok(ctx, pk, stm, batch, d_key_idx, d_st_idx) || !d_cipher || !proof || !d_ok) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  return mpe::alice_verify(ctx, pk, stm, batch, d_key_idx, d_st_idx, mpe::rows(d_cipher, 128), mpe::dense(proof), d_ok,
                           (hipStream_t)stream);
}
int mpe_pdl_prove(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch, const int32_t* d_key_idx,
                  const int32_t* d_st_idx, const uint32_t* d_cipher, const uint32_t* d_Q, const uint32_t* d_G,
                  const uint32_t* d_x, const uint32_t* d_r, const mpe_pdl_nonces* nonces, const mpe_pdl_proof* out,
                  void* stream) {
  if (!proof_args_ok(ctx, pk, stm, batch, d_key_idx, d_st_idx) || !d_cipher || !d_Q || !d_G || !d_x || !d_r || !nonces || !out)
    return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  return mpe::pdl_prove(ctx, pk, stm, batch, d_key_idx, d_st_idx, mpe::rows(d_cipher, 128), mpe::rows(d_Q, 16), mpe::rows(d_G, 16),
                        mpe::rows(d_x, 8), mpe::rows(d_r, 64), nonces, out, (hipStream_t)stream);
}
int mpe_pdl_verify(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch, const int32_t* d_key_idx,
                   const int32_t* d_st_idx, const uint32_t* d_cipher, const uint32_t* d_Q, const uint32_t* d_G,
                   const mpe_pdl_proof* proof, uint8_t* d_ok, void* stream) {
  if (!proof_args_ok(ctx, pk, stm, batch, d_key_idx, d_st_idx) || !d_cipher || !d_Q || !d_G || !proof || !d_ok) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  return mpe::pdl_verify(ctx, pk, stm, batch, d_key_idx, d_st_idx, mpe::rows(d_cipher, 128), mpe::rows(d_Q, 16), mpe::rows(d_G, 16),
                         mpe::dense(proof), d_ok, (hipStream_t)stream);
}

}  // extern "C"
