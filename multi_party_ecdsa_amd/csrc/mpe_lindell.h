// Lindell'17 two-party ECDSA, the signing half, batched over independent sessions:
//   party two  PartialSig::compute            src/protocols/two_party_ecdsa/lindell_2017/party_two.rs:390-423
//   party one  Signature::compute_with_recid  src/protocols/two_party_ecdsa/lindell_2017/party_one.rs:519-565
// Both are the Paillier kernels of the GG20 path plus two per-lane EC / scalar kernels: the partial signature
// c3 = Enc(rho q + k2^-1 m; r) * c_key^(k2^-1 rx x2) is ONE two-base ladder modulo N^2 (the peer's key: no p, q),
// party one's s = Dec(c3) k1^-1 is the CRT decryption on the pair kernel modulo p^2 | q^2.
// Key generation, the PDL exchange: party one `pdl_proof` (party_one.rs:366-401) and party two `PaillierPublic::pdl_verify`
// (party_two.rs:275-300) are compositions of PDLwSlackProof::{prove, verify} and CompositeDLogProof::verify over a
// statement (N~, h1, h2) that is fresh for every key: no fixed-base tables, every item its own moduli.
// Included by mpe_lib.hip (after mpe_proofs.h and mpe_keygen.h).
#pragma once
#include "mpe_paillier.h"
#include "mpe_ec.h"

namespace mpe {

// per session: plaintext ps = rho q + (k2^-1 m mod q)  [64 words]  and multiplier v = k2^-1 (rx x2) mod q  [8 words]
__global__ void __launch_bounds__(64) MPE_EC_OCC lindell_p2_prep_kernel(int B, const uint32_t* __restrict__ k2, const uint32_t* __restrict__ x2,
                                                             const uint32_t* __restrict__ R1, const uint32_t* __restrict__ msg,
                                                             const uint32_t* __restrict__ rho, uint32_t* __restrict__ ps,
                                                             uint32_t* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 k = ec::sc_reduce(k2 + (size_t)i * 8, 8);
  const ec::Aff R1p = ec::aff_load(R1 + (size_t)i * 16);
  if (!ec::aff_valid(R1p)) {      // the peer's ephemeral point is not on the curve (curv would not have deserialised it): fail closed,
    for (int j = 0; j < 64; ++j) ps[(size_t)i * 64 + j] = 0u;      // the secret k2 never touches it and c3 becomes Enc(0; r)
    for (int j = 0; j < 8; ++j) v[(size_t)i * 8 + j] = 0u;
    return;
  }
  const ec::Aff Rp = ec::jac_to_aff(ec::jac_mul(k, R1p));                                  // r = R1 * k2        :401
  const ec::U256 rx = ec::sc_reduce(Rp.x.w, 8);                                            // rx = r.x mod q     :403
  const ec::U256 kinv = ec::sc_inv(k);                                                     // k2_inv             :405
  const ec::U256 t = ec::sc_mul(kinv, ec::sc_reduce(msg + (size_t)i * 8, 8));
  uint32_t prod[24], q8[8], r16[16];
  for (int j = 0; j < 8; ++j) q8[j] = ec::FQ[j];
  sm::copy(r16, rho + (size_t)i * 16, 16);
  sm::mul(prod, r16, 16, q8, 8);                                                           // rho q
  sm::add(prod, 24, prod, 24, t.w, 8);                                                     // + k2_inv m         :406
  for (int j = 0; j < 64; ++j) ps[(size_t)i * 64 + j] = j < 24 ? prod[j] : 0u;
  const ec::U256 vv = ec::sc_mul(kinv, ec::sc_mul(rx, ec::sc_reduce(x2 + (size_t)i * 8, 8)));   // :409-413
  ec::u256_store(v + (size_t)i * 8, vv);
}

// per session: r = (k1 R2).x mod q, s = min(s'', q - s'') with s'' = (s_tag mod q) k1^-1, recid   (:526-561)
__global__ void __launch_bounds__(64) MPE_EC_OCC lindell_p1_finish_kernel(int B, const uint32_t* __restrict__ s_tag, const uint32_t* __restrict__ k1,
                                                               const uint32_t* __restrict__ R2, uint32_t* __restrict__ r_out,
                                                               uint32_t* __restrict__ s_out, int32_t* __restrict__ recid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 k = ec::sc_reduce(k1 + (size_t)i * 8, 8);
  const ec::Aff R2p = ec::aff_load(R2 + (size_t)i * 16);
  if (!ec::aff_valid(R2p)) {      // fail closed: no signature leaves, k1 never touches the point
    for (int j = 0; j < 8; ++j) { r_out[(size_t)i * 8 + j] = 0u; s_out[(size_t)i * 8 + j] = 0u; }
    recid[i] = -1;
    return;
  }
  const ec::Aff Rp = ec::jac_to_aff(ec::jac_mul(k, R2p));
  const ec::U256 rx = ec::sc_reduce(Rp.x.w, 8), ry = ec::sc_reduce(Rp.y.w, 8);
  ec::U256 s = ec::sc_mul(ec::sc_reduce(s_tag + (size_t)i * 64, 64), ec::sc_inv(k));
  const ec::U256 neg = ec::sc_neg(s);
  bool gt = false;
  for (int j = 7; j >= 0; --j) { if (s.w[j] != neg.w[j]) { gt = s.w[j] > neg.w[j]; break; } }
  int rec = (int)(ry.w[0] & 1u);
  if (gt) { s = neg; rec ^= 1; }
  ec::u256_store(r_out + (size_t)i * 8, rx);
  ec::u256_store(s_out + (size_t)i * 8, s);
  recid[i] = rec;
}

__global__ void lindell_gen_point_kernel(uint32_t* __restrict__ g) {
  if (threadIdx.x < 8) { g[threadIdx.x] = ec::GX[threadIdx.x]; g[8 + threadIdx.x] = ec::GY[threadIdx.x]; }
}
// ok[i] &= (statement's ek, ciphertext, Q) == (party two's ek, encrypted_secret_share, q1)          party_two.rs:282-288
__global__ void lindell_stmt_check_kernel(int B, const uint32_t* __restrict__ sN, const uint32_t* __restrict__ N, const int32_t* __restrict__ key_idx,
                                          const uint32_t* __restrict__ sc, const uint32_t* __restrict__ c, const uint32_t* __restrict__ sQ,
                                          const uint32_t* __restrict__ q1, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const uint32_t* Nk = N + (size_t)(key_idx ? key_idx[i] : i) * 64;
  uint32_t d = 0;
  for (int j = 0; j < 64; ++j) d |= sN[(size_t)i * 64 + j] ^ Nk[j];
  for (int j = 0; j < 128; ++j) d |= sc[(size_t)i * 128 + j] ^ c[(size_t)i * 128 + j];
  for (int j = 0; j < 16; ++j) d |= sQ[(size_t)i * 16 + j] ^ q1[(size_t)i * 16 + j];
  if (d) ok[i] = 0;
}
__global__ void and_u8_kernel(int B, uint8_t* __restrict__ a, const uint8_t* __restrict__ b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) a[i] = a[i] && b[i];
}

}  // namespace mpe

extern "C" {

int mpe_lindell_pdl_proof(mpe_ctx* ctx, const mpe_paillier* sk, const mpe_statements* stm, int batch, const int32_t* d_key_idx,
                          const int32_t* d_st_idx, const uint32_t* d_c_key, const uint32_t* d_x1, const uint32_t* d_r,
                          const mpe_pdl_nonces* nonces, uint32_t* d_Q, const mpe_pdl_proof* out, void* stream) {
  if (!ctx || !sk || !stm || !d_c_key || !d_x1 || !d_r || !nonces || !d_Q || !out || batch < 0) return MPE_E_ARG;
  if (!sk->has_private) { mpe_set_error_msg("mpe_lindell_pdl_proof: key set has no private part"); return MPE_E_ARG; }
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  uint32_t* gen = nullptr;
  if (hipMalloc((void**)&gen, 64) != hipSuccess) return MPE_E_NOMEM;
  hipLaunchKernelGGL(mpe::lindell_gen_point_kernel, dim3(1), dim3(64), 0, st, gen);
  hipLaunchKernelGGL(mpe::ec_mul_kernel, dim3(mpe::blocks_for(batch, 64)), dim3(64), 0, st, batch, d_x1, 8, (const uint32_t*)nullptr, d_Q);   // Q = x1 G  :384
  const int rc = mpe::pdl_prove(ctx, sk, stm, batch, d_key_idx, d_st_idx, mpe::rows(d_c_key, 128), mpe::rows(d_Q, 16), mpe::rows(gen, 0),
                                mpe::rows(d_x1, 8), mpe::rows(d_r, 64), nonces, out, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(gen);
  return rc;
}

int mpe_lindell_pdl_verify(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx, const uint32_t* d_Nt,
                           const uint32_t* d_h1, const uint32_t* d_h2, const uint32_t* d_dlog_x, const uint32_t* d_dlog_y,
                           const uint32_t* d_stmt_N, const uint32_t* d_stmt_c, const uint32_t* d_stmt_Q, const uint32_t* d_c_key,
                           const uint32_t* d_q1, const mpe_pdl_proof* proof, uint8_t* d_ok, void* stream) {
  if (!ctx || !pk || !d_Nt || !d_h1 || !d_h2 || !d_dlog_x || !d_dlog_y || !d_stmt_N || !d_stmt_c || !d_stmt_Q || !d_c_key || !d_q1 ||
      !proof || !d_ok || batch < 0)
    return MPE_E_ARG;
  if (!d_key_idx && pk->nkeys != 1 && pk->nkeys < batch) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  mpe_statements* stm = nullptr;
  MPE_TRY(mpe_statements_create_wb(ctx, batch, d_Nt, d_h1, d_h2, 0, &stm, stream));                        // statement i for item i
  uint8_t* ok2 = nullptr;
  if (hipMalloc((void**)&ok2, (size_t)batch + 64) != hipSuccess) { mpe_statements_destroy(stm); return MPE_E_NOMEM; }
  uint32_t* genp = nullptr;
  int rc = MPE_OK;
  if (hipMalloc((void**)&genp, 64) != hipSuccess) rc = MPE_E_NOMEM;
  if (rc == MPE_OK) {
    hipLaunchKernelGGL(mpe::lindell_gen_point_kernel, dim3(1), dim3(64), 0, st, genp);
    rc = mpe::pdl_verify(ctx, pk, stm, batch, d_key_idx, nullptr, mpe::rows(d_stmt_c, 128), mpe::rows(d_stmt_Q, 16), mpe::rows(genp, 0),
                         mpe::dense(proof), d_ok, st);                                                        // :297
  }
  if (rc == MPE_OK) rc = mpe_composite_dlog_verify(ctx, batch, d_Nt, d_h1, d_h2, d_dlog_x, d_dlog_y, ok2, stream);   // :296
  if (rc == MPE_OK) {
    hipLaunchKernelGGL(mpe::and_u8_kernel, dim3(mpe::blocks_for(batch, 64)), dim3(64), 0, st, batch, d_ok, ok2);
    hipLaunchKernelGGL(mpe::lindell_stmt_check_kernel, dim3(mpe::blocks_for(batch, 64)), dim3(64), 0, st, batch, d_stmt_N, pk->N, d_key_idx,
                       d_stmt_c, d_c_key, d_stmt_Q, d_q1, d_ok);
    if (hipGetLastError() != hipSuccess) rc = MPE_E_HIP;
  }
  (void)hipStreamSynchronize(st);
  if (genp) (void)hipFree(genp);
  (void)hipFree(ok2);
  mpe_statements_destroy(stm);
  return rc;
}

int mpe_lindell_partial_sig(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx, const uint32_t* d_c_key,
                            const uint32_t* d_x2, const uint32_t* d_k2, const uint32_t* d_R1, const uint32_t* d_msg,
                            const uint32_t* d_rho, const uint32_t* d_r, uint32_t* d_c3, void* stream) {
  if (!ctx || !pk || !d_c_key || !d_x2 || !d_k2 || !d_R1 || !d_msg || !d_rho || !d_r || !d_c3 || batch < 0) return MPE_E_ARG;
  if (!d_key_idx && pk->nkeys != 1 && pk->nkeys < batch) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  // own arrays at the top of the workspace, the composite below (the scheme of mpe_mta.h)
  MPE_TRY(mpe::ws_reserve(ctx, (size_t)batch * (128 * 3 + 64 + 8 + 64) * 4 + (1u << 20), st));
  char* top = (char*)ctx->ws + ctx->ws_bytes;
  uint32_t* ps = (uint32_t*)(top -= ((size_t)batch * 64 * 4 + 255) & ~(size_t)255);
  uint32_t* v = (uint32_t*)(top -= ((size_t)batch * 8 * 4 + 255) & ~(size_t)255);
  mpe::WsTop hold(ctx, top);
  MPE_LAUNCH_1D(mpe::lindell_p2_prep_kernel, batch, st, batch, d_k2, d_x2, d_R1, d_msg, d_rho, ps, v);
  // c3 = c_key^v * Enc(ps; r): Paillier::encrypt, Paillier::mul, Paillier::add   :408-421
  return mpe::paillier_mul_add_enc(ctx, pk, batch, d_key_idx, mpe::rows(d_c_key, 128), mpe::rows(v, 8), 8, ps, d_r, d_c3, st);
}

int mpe_lindell_sign(mpe_ctx* ctx, const mpe_paillier* sk, int batch, const int32_t* d_key_idx, const uint32_t* d_c3,
                     const uint32_t* d_k1, const uint32_t* d_R2, uint32_t* d_r, uint32_t* d_s, int32_t* d_recid, void* stream) {
  if (!ctx || !sk || !d_c3 || !d_k1 || !d_R2 || !d_r || !d_s || !d_recid || batch < 0) return MPE_E_ARG;
  if (!sk->has_private) { mpe_set_error_msg("mpe_lindell_sign: key set has no private part"); return MPE_E_ARG; }
  if (!d_key_idx && sk->nkeys != 1 && sk->nkeys < batch) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  MPE_TRY(mpe::ws_reserve(ctx, (size_t)batch * (2 * (3 + 64 + 32 + 64 + 64) + 64) * 4 + (1u << 20), st));
  char* top = (char*)ctx->ws + ctx->ws_bytes;
  uint32_t* s_tag = (uint32_t*)(top -= ((size_t)batch * 64 * 4 + 255) & ~(size_t)255);
  mpe::WsTop hold(ctx, top);
  MPE_TRY(mpe::paillier_decrypt(ctx, sk, batch, d_key_idx, mpe::rows(d_c3, 128), s_tag, st));               // :538-542
  MPE_LAUNCH_1D(mpe::lindell_p1_finish_kernel, batch, st, batch, s_tag, d_k1, d_R2, d_r, d_s, d_recid);
  return MPE_OK;
}

}  // extern "C"
