// Bob's MtA(wc) range proof on the GPU:
//   BobProof::generate / verify, BobProofExt::verify     src/utilities/mta/range_proofs.rs:218-534
// (in north_star scope; not called by the GG20 state machine itself — SURVEY.md §8a row a24).
// Same construction as mpe_proofs.h: host-sequenced batched launches, intermediates in the workspace.
#pragma once
#include "mpe_proofs.h"

namespace mpe {

// out = (a + 1) mod m   (a < m; used for (gamma N + 1) mod N^2 where gamma N may exceed N^2)
template <int K32>
__global__ void add_one_mod_kernel(int B, const uint32_t* __restrict__ a, const uint32_t* __restrict__ mod_words, Rows sel,
                                   uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t x[K32 + 1];
  const uint32_t one[1] = {1};
  x[K32] = sm::add(x, K32, a + (size_t)i * K32, K32, one, 1);
  const uint32_t* m = mod_words + (size_t)mod_index(sel, i) * K32;
  if (sm::cmp(x, K32 + 1, m, K32) >= 0) sm::sub(x, K32 + 1, x, K32 + 1, m, K32);
  sm::copy(out + (size_t)i * K32, x, K32);
}
// ok &= ( (s1 mod q) G == (e mod q) X + u )        BobProofExt::verify :522-531
__global__ void __launch_bounds__(64) MPE_EC_OCC bob_ext_check_kernel(int B, Rows s1, Rows e, const uint32_t* __restrict__ X, const uint32_t* __restrict__ u,
                                     uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 a = ec::sc_reduce(row_of(s1, i), 25), ee = ec::sc_reduce(row_of(e, i), 8);
  const ec::Aff Xp = ec::aff_load(X + (size_t)i * 16), up = ec::aff_load(u + (size_t)i * 16);
  if (!ec::aff_valid(Xp) || !ec::aff_valid(up)) { ok[i] = 0; return; }
  const ec::Jac l = ec::jac_mul_gen(a);
  const ec::Jac r = ec::jac_add_aff(ec::jac_mul(ee, Xp), up);
  if (!ec::jac_eq(l, r)) ok[i] = 0;
}

struct BobProofRows { Rows t, z, e, s, s1, s2, t1, t2; };
static BobProofRows dense(const mpe_bob_proof* p) {
  return BobProofRows{rows(p->t, 64), rows(p->z, 64), rows(p->e, 8), rows(p->s, 64), rows(p->s1, 25), rows(p->s2, 89),
                      rows(p->t1, 81), rows(p->t2, 89)};
}

static void bob_hash_desc(HashDesc& d, Rows Nrow, Rows a_enc, Rows mta, Rows z, Rows zp, Rows t, Rows v, Rows w,
                          const uint32_t* X, const uint32_t* u) {
  d.n = 9;
  d.f[0] = hf(Nrow, 64); d.f[1] = hf(Nrow, 64, HF_BIGINT_PLUS1); d.f[2] = hf(a_enc, 128); d.f[3] = hf(mta, 128);
  d.f[4] = hf(z, 64); d.f[5] = hf(zp, 64); d.f[6] = hf(t, 64); d.f[7] = hf(v, 128); d.f[8] = hf(w, 64);
  if (X && u) {                              // x_coord / y_coord as BigInt  (range_proofs.rs:375-405,446-470)
    d.f[9] = hf(rows(X, 16), 8); d.f[10] = hf(rows(X + 8, 16), 8); d.f[11] = hf(rows(u, 16), 8); d.f[12] = hf(rows(u + 8, 16), 8);
    d.n = 13;
  }
}

// BobProof::generate (range_proofs.rs:414-487; BobZkpRound1 :218-264, BobZkpRound2 :281-297)
static int bob_generate(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int B, const int32_t* key_idx,
                        const int32_t* st_idx, const uint32_t* a_enc, const uint32_t* mta_enc, const uint32_t* b,
                        const uint32_t* beta_prim, const uint32_t* r, const mpe_bob_nonces* nn, int check,
                        const mpe_bob_proof* out, uint32_t* u_out, hipStream_t st) {
  MPE_TRY(ws_reserve(ctx, (size_t)B * 2400 * 4 + 65536, st));
  Seq q{ctx, st, B};
  const Rows ksel = sel_of(key_idx, pk->nkeys), ssel = sel_of(st_idx, stm->count);
  const Rows h1 = tab_rows(stm->h1, 64, st_idx, stm->count), h2 = tab_rows(stm->h2, 64, st_idx, stm->count);
  const Rows Nrow = tab_rows(pk->N, 64, key_idx, pk->nkeys);
  auto commit = [&](Rows x, int xw, Rows y, int yw, uint32_t* dst) {      // h1^x h2^y mod N~
    uint32_t* p1 = q.fb_modexp(stm, ssel, 0, h1, x, xw);
    uint32_t* p2 = q.fb_modexp(stm, ssel, 1, h2, y, yw);
    q.modmul_to(stm->ms, ssel, rows(p1, 64), rows(p2, 64), dst);
  };
  uint32_t *zp = q.words(64), *w = q.words(64);
  commit(rows(b, 8), 8, rows(nn->rho, 72), 72, out->z);                                 // z       :238
  commit(rows(nn->alpha, 24), 24, rows(nn->rho_prim, 88), 88, zp);                       // z'      :239-241
  commit(rows(beta_prim, 64), 64, rows(nn->sigma, 72), 72, out->t);                      // t       :242-243
  commit(rows(nn->gamma, 80), 80, rows(nn->tau, 88), 88, w);                             // w       :244-245
  // v = a_enc^alpha (gamma N + 1) beta^N mod N^2                                          :246-249
  uint32_t* ca = q.modexp_nn(pk, ksel, rows(a_enc, 128), rows(nn->alpha, 24), 24, false);      // Bob works under Alice's key
  uint32_t* gN = q.modmul(pk->ms_nn, ksel, rows(nn->gamma, 80, nullptr, 80), with_words(Nrow, 64));
  uint32_t* g1 = q.words(128);
  if (q.rc == MPE_OK)
    hipLaunchKernelGGL(add_one_mod_kernel<128>, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, gN, pk->ms_nn->words, ksel, g1);
  uint32_t* bn = q.modexp_nn(pk, ksel, rows(nn->beta, 64, nullptr, 64), Nrow, 64, false);
  uint32_t* v1 = q.modmul(pk->ms_nn, ksel, rows(ca, 128), rows(g1, 128));
  uint32_t* v = q.modmul(pk->ms_nn, ksel, rows(v1, 128), rows(bn, 128));
  // e = H(N, N+1, a_enc, mta, z, z', t, v, w [, X.x, X.y, u.x, u.y])                       :433-470
  uint32_t* X = nullptr;
  if (check) {
    X = q.words(16);
    if (q.rc == MPE_OK) {
      hipLaunchKernelGGL(ec_mul_rows_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, rows(b, 8), 8, no_rows(), X);
      hipLaunchKernelGGL(ec_mul_rows_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, rows(nn->alpha, 24), 24, no_rows(), u_out);
    }
  }
  HashDesc d;
  bob_hash_desc(d, Nrow, rows(a_enc, 128), rows(mta_enc, 128), rows(out->z, 64), rows(zp, 64), rows(out->t, 64), rows(v, 128),
                rows(w, 64), check ? X : nullptr, check ? u_out : nullptr);
  q.hash(d, out->e);
  // round 2: s = r^e beta mod N; s1 = e b + alpha; s2 = e rho + rho'; t1 = e beta' + gamma; t2 = e sigma + tau   :290-296
  uint32_t* re = q.modexp(pk->ms_n, ksel, rows(r, 64), rows(out->e, 8), 8);
  q.modmul_to(pk->ms_n, ksel, rows(re, 64), rows(nn->beta, 64), out->s);
  q.muladd(rows(out->e, 8), 8, rows(b, 8), 8, rows(nn->alpha, 24), 24, out->s1, 25);
  q.muladd(rows(out->e, 8), 8, rows(nn->rho, 72), 72, rows(nn->rho_prim, 88), 88, out->s2, 89);
  q.muladd(rows(out->e, 8), 8, rows(beta_prim, 64), 64, rows(nn->gamma, 80), 80, out->t1, 81);
  q.muladd(rows(out->e, 8), 8, rows(nn->sigma, 72), 72, rows(nn->tau, 88), 88, out->t2, 89);
  return q.finish("bob_generate");
}

// BobProof::verify (range_proofs.rs:321-412); X,u non-null -> BobProofExt::verify (:499-534)
static int bob_verify(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int B, const int32_t* key_idx,
                      const int32_t* st_idx, const uint32_t* a_enc, const uint32_t* mta_enc, const BobProofRows& pr,
                      const uint32_t* X, const uint32_t* u, uint8_t* ok, hipStream_t st) {
  MPE_TRY(ws_reserve(ctx, (size_t)B * (3600 + 3 * CRT_WS_WORDS) * 4 + 65536, st));
  Seq q{ctx, st, B};
  const Rows ksel = sel_of(key_idx, pk->nkeys), ssel = sel_of(st_idx, stm->count);
  const Rows h1 = tab_rows(stm->h1, 64, st_idx, stm->count), h2 = tab_rows(stm->h2, 64, st_idx, stm->count);
  const Rows Nrow = tab_rows(pk->N, 64, key_idx, pk->nkeys);
  MPE_LAUNCH_1D(s1_range_kernel, B, st, B, pr.s1, 25, ok);                                            // :335
  uint8_t *ok1 = q.flags(), *ok2 = q.flags(), *ok3 = q.flags();
  auto open = [&](Rows x, int xw, Rows y, int yw, Rows c, uint8_t* okf) -> uint32_t* {               // h1^x h2^y (c^e)^-1 mod N~
    uint32_t* ce = q.modexp(stm->ms, ssel, c, pr.e, 8);
    uint32_t* cei = q.modinv(stm->ms, ssel, rows(ce, 64), okf);
    uint32_t* p1 = q.fb_modexp(stm, ssel, 0, h1, x, xw);
    uint32_t* p2 = q.fb_modexp(stm, ssel, 1, h2, y, yw);
    uint32_t* p12 = q.modmul(stm->ms, ssel, rows(p1, 64), rows(p2, 64));
    return q.modmul(stm->ms, ssel, rows(p12, 64), rows(cei, 64));
  };
  uint32_t* zp = open(pr.s1, 25, pr.s2, 89, pr.z, ok1);                                               // z'  :339-349
  uint32_t* w = open(pr.t1, 81, pr.t2, 89, pr.t, ok3);                                                // w   :363-372
  // v = a_enc^s1 s^N (t1 N + 1) (mta^e)^-1 mod N^2                                                     :351-361
  // the verifier of a Bob proof is Alice, the owner of the key: her exponentiations may go through p^2 | q^2
  uint32_t* me = q.modexp_nn(pk, ksel, rows(mta_enc, 128), pr.e, 8, true);
  uint32_t* mei = q.modinv(pk->ms_nn, ksel, rows(me, 128), ok2);
  uint32_t* as1 = q.modexp_nn(pk, ksel, rows(a_enc, 128), pr.s1, 25, true);
  uint32_t* sn = q.modexp_nn(pk, ksel, with_words(pr.s, 64), Nrow, 64, true, true);
  uint32_t* tN = q.modmul(pk->ms_nn, ksel, with_words(pr.t1, 81), with_words(Nrow, 64));
  uint32_t* g1 = q.words(128);
  if (q.rc == MPE_OK)
    hipLaunchKernelGGL(add_one_mod_kernel<128>, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, tN, pk->ms_nn->words, ksel, g1);
  uint32_t* v1 = q.modmul(pk->ms_nn, ksel, rows(as1, 128), rows(sn, 128));
  uint32_t* v2 = q.modmul(pk->ms_nn, ksel, rows(v1, 128), rows(g1, 128));
  uint32_t* v = q.modmul(pk->ms_nn, ksel, rows(v2, 128), rows(mei, 128));
  uint32_t* e2 = q.words(8);
  HashDesc d;
  bob_hash_desc(d, Nrow, rows(a_enc, 128), rows(mta_enc, 128), pr.z, rows(zp, 64), pr.t, rows(v, 128), rows(w, 64), X, u);
  q.hash(d, e2);
  if (q.rc == MPE_OK) {
    hipLaunchKernelGGL(and_flags_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, ok, ok1, ok2, e2, pr.e, 8);
    hipLaunchKernelGGL(and_flags_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, ok, ok3, (const uint8_t*)nullptr,
                       (const uint32_t*)nullptr, no_rows(), 0);
    if (X && u) hipLaunchKernelGGL(bob_ext_check_kernel, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, pr.s1, pr.e, X, u, ok);
  }
  return q.finish("bob_verify");
}

}  // namespace mpe

extern "C" {

int mpe_bob_generate(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch, const int32_t* d_key_idx,
                     const int32_t* d_st_idx, const uint32_t* d_a_enc, const uint32_t* d_mta_enc, const uint32_t* d_b,
                     const uint32_t* d_beta_prim, const uint32_t* d_r, const mpe_bob_nonces* nonces, int check,
                     const mpe_bob_proof* out, uint32_t* d_u, void* stream) {
  if (!proof_args_ok(ctx, pk, stm, batch, d_key_idx, d_st_idx) || !d_a_enc || !d_mta_enc || !d_b || !d_beta_prim || !d_r ||
      !nonces || !out || (check && !d_u))
    return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  return mpe::bob_generate(ctx, pk, stm, batch, d_key_idx, d_st_idx, d_a_enc, d_mta_enc, d_b, d_beta_prim, d_r, nonces, check, out,
                           d_u, (hipStream_t)stream);
}
int mpe_bob_verify(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch, const int32_t* d_key_idx,
                   const int32_t* d_st_idx, const uint32_t* d_a_enc, const uint32_t* d_mta_enc, const mpe_bob_proof* proof,
                   const uint32_t* d_X, const uint32_t* d_u, uint8_t* d_ok, void* stream) {
  if (!proof_args_ok(ctx, pk, stm, batch, d_key_idx, d_st_idx) || !d_a_enc || !d_mta_enc || !proof || !d_ok ||
      ((d_X == nullptr) != (d_u == nullptr)))
    return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  return mpe::bob_verify(ctx, pk, stm, batch, d_key_idx, d_st_idx, d_a_enc, d_mta_enc, mpe::dense(proof), d_X, d_u, d_ok,
                         (hipStream_t)stream);
}

}  // extern "C"
