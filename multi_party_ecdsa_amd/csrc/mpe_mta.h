// MtA share conversion, the three calls GG20's Round0/1/2 make (src/utilities/mta/mod.rs):
//   MessageA::a_with_predefined_randomness   :62-87    -> mpe_mta_message_a
//   MessageB::b_with_predefined_randomness   :111-158   -> mpe_mta_message_b
//   MessageB::verify_proofs_get_alpha        :160-179   -> mpe_mta_verify_get_alpha
// Batched over B independent (Alice, Bob) exchanges; `dlog_statements` = all `count` statements of the
// statement set (rounds.rs:87,154 pass the whole h1_h2_n_tilde_vec).  Included by mpe_lib.hip.
#pragma once
#include "mpe_proofs.h"

namespace mpe {

// item g = b * nst + st  ->  (b, st), key of b
__global__ void mta_idx_kernel(int total, int nst, int nkeys, const int32_t* __restrict__ key_idx, int32_t* __restrict__ b_of,
                               int32_t* __restrict__ st_of, int32_t* __restrict__ key_of_item) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const int b = g / nst;
  b_of[g] = b;
  st_of[g] = g - b * nst;
  key_of_item[g] = key_of(key_idx, nkeys, b);
}
// ok[b] = AND over the nst statements
__global__ void mta_all_kernel(int B, int nst, const uint8_t* __restrict__ ok_items, uint8_t* __restrict__ ok) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  bool v = true;
  for (int s = 0; s < nst; ++s) v = v && ok_items[(size_t)b * nst + s];
  ok[b] = v ? 1 : 0;
}
// beta_tag_fe = beta_tag mod q ; beta = -beta_tag_fe            (mta/mod.rs:132,146)
__global__ void mta_beta_kernel(int B, const uint32_t* __restrict__ beta_tag, uint32_t* __restrict__ btq, uint32_t* __restrict__ beta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 t = ec::sc_reduce(beta_tag + (size_t)i * 64, 64);
  ec::u256_store(btq + (size_t)i * 8, t);
  ec::u256_store(beta + (size_t)i * 8, ec::sc_neg(t));
}
// alpha = alice_share mod q; ok = DLogProof::verify x2 && b_proof.pk * a + beta_tag_proof.pk == g^alpha   (:166-178)
__global__ void __launch_bounds__(64) MPE_EC_OCC mta_alpha_kernel(int B, ec::Enc enc, const uint32_t* __restrict__ share, const uint32_t* __restrict__ a,
                                 const uint32_t* __restrict__ pk, const uint32_t* __restrict__ R, const uint32_t* __restrict__ z,
                                 const uint32_t* __restrict__ tpk, const uint32_t* __restrict__ tR, const uint32_t* __restrict__ tz,
                                 uint32_t* __restrict__ alpha, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 al = ec::sc_reduce(share + (size_t)i * 64, 64);
  ec::u256_store(alpha + (size_t)i * 8, al);
  const ec::Aff Bpk = ec::aff_load(pk + (size_t)i * 16), BTpk = ec::aff_load(tpk + (size_t)i * 16);
  const ec::Aff R1 = ec::aff_load(R + (size_t)i * 16), R2 = ec::aff_load(tR + (size_t)i * 16);
  if (!ec::aff_valid(Bpk) || !ec::aff_valid(BTpk) || !ec::aff_valid(R1) || !ec::aff_valid(R2)) { ok[i] = 0; return; }   // the secret a is never multiplied into an unchecked point
  const ec::Jac g_alpha = ec::jac_mul_gen(al);
  const ec::Jac bb = ec::jac_add_aff(ec::jac_mul(ec::sc_reduce(a + (size_t)i * 8, 8), Bpk), BTpk);
  bool good = ec::jac_eq(g_alpha, bb);
  const ec::U256 c1 = dlog_challenge(R1, Bpk, enc), c2 = dlog_challenge(R2, BTpk, enc);
  const ec::Jac l1 = ec::jac_add(ec::jac_mul_gen(ec::sc_reduce(z + (size_t)i * 8, 8)), ec::jac_mul(c1, Bpk));
  const ec::Jac l2 = ec::jac_add(ec::jac_mul_gen(ec::sc_reduce(tz + (size_t)i * 8, 8)), ec::jac_mul(c2, BTpk));
  good = good && ec::jac_eq_aff(l1, R1) && ec::jac_eq_aff(l2, R2);
  ok[i] = good ? 1 : 0;
}

}  // namespace mpe

extern "C" {

int mpe_mta_message_a(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch, const int32_t* d_key_idx,
                      const uint32_t* d_a, const uint32_t* d_r, const mpe_alice_nonces* nonces, uint32_t* d_c,
                      const mpe_alice_proof* proofs, void* stream) {
  if (!ctx || !pk || !stm || !d_a || !d_r || !nonces || !d_c || !proofs || batch < 0) return MPE_E_ARG;
  if (!d_key_idx && pk->nkeys != 1 && pk->nkeys < batch) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nst = stm->count, total = batch * nst;
  // own arrays at the top of the workspace, composites below (same scheme as the GG20 pipeline)
  MPE_TRY(mpe::ws_reserve(ctx, ((size_t)total * 2100 + (size_t)batch * 1000) * 4 + (1u << 20), st));
  char* top = (char*)ctx->ws + ctx->ws_bytes;
  int32_t* b_of = (int32_t*)(top -= ((size_t)total * 4 + 255) & ~(size_t)255);
  int32_t* st_of = (int32_t*)(top -= ((size_t)total * 4 + 255) & ~(size_t)255);
  int32_t* key_it = (int32_t*)(top -= ((size_t)total * 4 + 255) & ~(size_t)255);
  uint32_t* a64 = (uint32_t*)(top -= ((size_t)batch * 64 * 4 + 255) & ~(size_t)255);
  mpe::WsTop hold(ctx, top);
  MPE_LAUNCH_1D(mpe::mta_idx_kernel, total, st, total, nst, pk->nkeys, d_key_idx, b_of, st_of, key_it);
  // c = Enc(a; r)   (:68-75)   a zero-extended to the plaintext width
  (void)hipMemsetAsync(a64, 0, (size_t)batch * 64 * 4, st);
  (void)hipMemcpy2DAsync(a64, 64 * 4, d_a, 8 * 4, 8 * 4, batch, hipMemcpyDeviceToDevice, st);
  MPE_TRY(mpe::paillier_encrypt(ctx, pk, batch, d_key_idx, a64, d_r, d_c, true, st));      // Alice's own key
  // one AliceProof per statement   (:76-81)
  return mpe::alice_generate(ctx, pk, stm, total, key_it, st_of, mpe::rows(d_a, 8, b_of), mpe::rows(d_c, 128, b_of),
                             mpe::rows(d_r, 64, b_of), nonces, proofs, st);
}

int mpe_mta_message_b(mpe_ctx* ctx, const mpe_paillier* pk, const mpe_statements* stm, int batch, const int32_t* d_key_idx,
                      const uint32_t* d_b, const uint32_t* d_ca, const mpe_alice_proof* range_proofs, const uint32_t* d_r,
                      const uint32_t* d_beta_tag, const uint32_t* d_nonce_b, const uint32_t* d_nonce_bt, uint32_t* d_cb,
                      uint32_t* d_beta, const mpe_dlog_proof* b_proof, const mpe_dlog_proof* beta_tag_proof, uint8_t* d_ok,
                      void* stream) {
  if (!ctx || !pk || !stm || !d_b || !d_ca || !range_proofs || !d_r || !d_beta_tag || !d_nonce_b || !d_nonce_bt || !d_cb ||
      !d_beta || !b_proof || !beta_tag_proof || !d_ok || batch < 0)
    return MPE_E_ARG;
  if (!d_key_idx && pk->nkeys != 1 && pk->nkeys < batch) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nst = stm->count, total = batch * nst;
  MPE_TRY(mpe::ws_reserve(ctx, ((size_t)total * 2400 + (size_t)batch * 700) * 4 + (1u << 20), st));
  char* top = (char*)ctx->ws + ctx->ws_bytes;
  auto take = [&](size_t bytes) { top -= (bytes + 255) & ~(size_t)255; return (void*)top; };
  int32_t* b_of = (int32_t*)take((size_t)total * 4);
  int32_t* st_of = (int32_t*)take((size_t)total * 4);
  int32_t* key_it = (int32_t*)take((size_t)total * 4);
  uint8_t* ok_items = (uint8_t*)take((size_t)total);
  uint32_t* btq = (uint32_t*)take((size_t)batch * 8 * 4);
  mpe::WsTop hold(ctx, top);
  MPE_LAUNCH_1D(mpe::mta_idx_kernel, total, st, total, nst, pk->nkeys, d_key_idx, b_of, st_of, key_it);
  // verify Alice's range proofs against every statement   (:119-131); any failure -> Err(InvalidKey) -> ok = 0
  MPE_TRY(mpe::alice_verify(ctx, pk, stm, total, key_it, st_of, mpe::rows(d_ca, 128, b_of), mpe::dense(range_proofs), ok_items, st));
  MPE_LAUNCH_1D(mpe::mta_all_kernel, batch, st, batch, nst, ok_items, d_ok);
  // c_b = (b * c_a) + Enc(beta_tag; r)   (:133-145);  beta = -beta_tag mod q   (:146)
  MPE_TRY(mpe::paillier_mul_add_enc(ctx, pk, batch, d_key_idx, mpe::rows(d_ca, 128), mpe::rows(d_b, 8), 8, d_beta_tag, d_r, d_cb,
                                    st));                                                      // Alice's key, Bob computes
  MPE_LAUNCH_1D(mpe::mta_beta_kernel, batch, st, batch, d_beta_tag, btq, d_beta);
  // DLogProof::prove(b), DLogProof::prove(beta_tag_fe)   (:147-148)
  MPE_LAUNCH_1D(mpe::dlog_prove_kernel, batch, st, batch, ctx->enc, d_b, d_nonce_b, b_proof->pk, b_proof->R, b_proof->z);
  MPE_LAUNCH_1D(mpe::dlog_prove_kernel, batch, st, batch, ctx->enc, btq, d_nonce_bt, beta_tag_proof->pk, beta_tag_proof->R, beta_tag_proof->z);
  return MPE_OK;
}

int mpe_mta_verify_get_alpha(mpe_ctx* ctx, const mpe_paillier* sk, int batch, const int32_t* d_key_idx, const uint32_t* d_cb,
                             const mpe_dlog_proof* b_proof, const mpe_dlog_proof* beta_tag_proof, const uint32_t* d_a,
                             uint32_t* d_alpha, uint32_t* d_alice_share, uint8_t* d_ok, void* stream) {
  if (!ctx || !sk || !d_cb || !b_proof || !beta_tag_proof || !d_a || !d_alpha || !d_alice_share || !d_ok || batch < 0) return MPE_E_ARG;
  if (!sk->has_private) { mpe_set_error_msg("mpe_mta_verify_get_alpha: key set has no private part"); return MPE_E_ARG; }
  if (!d_key_idx && sk->nkeys != 1 && sk->nkeys < batch) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  MPE_TRY(mpe::paillier_decrypt(ctx, sk, batch, d_key_idx, mpe::rows(d_cb, 128), d_alice_share, st));             // :165
  MPE_LAUNCH_1D(mpe::mta_alpha_kernel, batch, st, batch, ctx->enc, d_alice_share, d_a, b_proof->pk, b_proof->R, b_proof->z,
                beta_tag_proof->pk, beta_tag_proof->R, beta_tag_proof->z, d_alpha, d_ok);
  return MPE_OK;
}

}  // extern "C"
