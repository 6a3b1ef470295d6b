// curv sigma proofs and the hash commitment GG20 uses in phases 1, 3 and 6 (un-vendored curv-kzen 0.9),
// as stand-alone batched entry points — the round pipeline (mpe_gg20.h) computes the same transcripts inline:
//   PedersenProof::{prove, verify}        party_i.rs:620-634, rounds.rs:371-378
//   HomoELGamalProof::{prove, verify}     party_i.rs:778-833
//   HashCommitment::create_commitment_with_user_defined_randomness   party_i.rs:577-580,654-659
// One item per lane.  Included by mpe_lib.hip.
#pragma once
#include "mpe_gg20.h"

namespace mpe {

__global__ void __launch_bounds__(64) MPE_EC_OCC pedersen_prove_kernel(int B, ec::Enc enc, const uint32_t* __restrict__ m, const uint32_t* __restrict__ r,
                                                            const uint32_t* __restrict__ s1_in, const uint32_t* __restrict__ s2_in,
                                                            mpe_pedersen_proof p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 mm = ec::sc_reduce(m + (size_t)i * 8, 8), rr = ec::sc_reduce(r + (size_t)i * 8, 8);
  const ec::U256 s1 = ec::sc_reduce(s1_in + (size_t)i * 8, 8), s2 = ec::sc_reduce(s2_in + (size_t)i * 8, 8);
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2();
  const ec::Aff C = ec::jac_to_aff(ec::jac_add(ec::jac_mul_gen(mm), ec::jac_mul_h2(rr)));
  const ec::Aff a1 = ec::jac_to_aff(ec::jac_mul_gen(s1)), a2 = ec::jac_to_aff(ec::jac_mul_h2(s2));
  const ec::Aff hp[5] = {G, H, C, a1, a2};
  const ec::U256 e = gg::hash_points(hp, enc, enc.ord_pedersen);
  ec::aff_store(p.com + (size_t)i * 16, C);
  ec::u256_store(p.e + (size_t)i * 8, e);
  ec::aff_store(p.a1 + (size_t)i * 16, a1);
  ec::aff_store(p.a2 + (size_t)i * 16, a2);
  ec::u256_store(p.z1 + (size_t)i * 8, ec::sc_add(s1, ec::sc_mul(e, mm)));
  ec::u256_store(p.z2 + (size_t)i * 8, ec::sc_add(s2, ec::sc_mul(e, rr)));
}
__global__ void __launch_bounds__(64) MPE_EC_OCC pedersen_verify_kernel(int B, ec::Enc enc, mpe_pedersen_proof p, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2();
  const ec::Aff C = ec::aff_load(p.com + (size_t)i * 16), a1 = ec::aff_load(p.a1 + (size_t)i * 16), a2 = ec::aff_load(p.a2 + (size_t)i * 16);
  if (!ec::aff_valid(C) || !ec::aff_valid(a1) || !ec::aff_valid(a2)) { ok[i] = 0; return; }
  const ec::Aff hp[5] = {G, H, C, a1, a2};
  const ec::U256 e = gg::hash_points(hp, enc, enc.ord_pedersen);
  const ec::Jac lhs = ec::jac_add(ec::jac_mul_gen(ec::sc_reduce(p.z1 + (size_t)i * 8, 8)), ec::jac_mul_h2(ec::sc_reduce(p.z2 + (size_t)i * 8, 8)));
  const ec::Jac rhs = ec::jac_add_aff(ec::jac_add_aff(ec::jac_mul(e, C), a1), a2);
  ok[i] = ec::jac_eq(lhs, rhs) ? 1 : 0;
}
__global__ void __launch_bounds__(64) MPE_EC_OCC heg_prove_kernel(int B, ec::Enc enc, const uint32_t* __restrict__ x, const uint32_t* __restrict__ r,
                                                       const uint32_t* __restrict__ s1_in, const uint32_t* __restrict__ s2_in,
                                                       mpe_heg_statement s, mpe_heg_proof p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::U256 xx = ec::sc_reduce(x + (size_t)i * 8, 8), rr = ec::sc_reduce(r + (size_t)i * 8, 8);
  const ec::U256 s1 = ec::sc_reduce(s1_in + (size_t)i * 8, 8), s2 = ec::sc_reduce(s2_in + (size_t)i * 8, 8);
  const ec::Aff G = ec::aff_load(s.G + (size_t)i * 16), H = ec::aff_load(s.H + (size_t)i * 16), Y = ec::aff_load(s.Y + (size_t)i * 16),
                D = ec::aff_load(s.D + (size_t)i * 16), E = ec::aff_load(s.E + (size_t)i * 16);
  const ec::Aff A3 = gg::mul_aff(s2, G), T = ec::jac_to_aff(ec::jac_add(ec::jac_mul(s1, H), ec::jac_mul(s2, Y)));
  const ec::Aff hp[7] = {T, A3, G, H, Y, D, E};
  const ec::U256 e = gg::hash_points(hp, enc, enc.ord_heg);
  ec::aff_store(p.T + (size_t)i * 16, T);
  ec::aff_store(p.A3 + (size_t)i * 16, A3);
  ec::u256_store(p.z1 + (size_t)i * 8, ec::u256_is_zero(xx) ? s1 : ec::sc_add(s1, ec::sc_mul(e, xx)));
  ec::u256_store(p.z2 + (size_t)i * 8, ec::sc_add(s2, ec::sc_mul(e, rr)));
}
__global__ void __launch_bounds__(64) MPE_EC_OCC heg_verify_kernel(int B, ec::Enc enc, mpe_heg_statement s, mpe_heg_proof p, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const ec::Aff G = ec::aff_load(s.G + (size_t)i * 16), H = ec::aff_load(s.H + (size_t)i * 16), Y = ec::aff_load(s.Y + (size_t)i * 16),
                D = ec::aff_load(s.D + (size_t)i * 16), E = ec::aff_load(s.E + (size_t)i * 16);
  const ec::Aff T = ec::aff_load(p.T + (size_t)i * 16), A3 = ec::aff_load(p.A3 + (size_t)i * 16);
  if (!(ec::aff_valid(G) && ec::aff_valid(H) && ec::aff_valid(Y) && ec::aff_valid(D) && ec::aff_valid(E) && ec::aff_valid(T) && ec::aff_valid(A3))) { ok[i] = 0; return; }
  const ec::Aff hp[7] = {T, A3, G, H, Y, D, E};
  const ec::U256 e = gg::hash_points(hp, enc, enc.ord_heg), z1 = ec::sc_reduce(p.z1 + (size_t)i * 8, 8), z2 = ec::sc_reduce(p.z2 + (size_t)i * 8, 8);
  const ec::Jac l1 = ec::jac_add(ec::jac_mul(z1, H), ec::jac_mul(z2, Y));
  const ec::Jac r1 = ec::jac_add_aff(ec::jac_mul(e, D), T);
  const ec::Jac l2 = ec::jac_mul(z2, G);
  const ec::Jac r2 = ec::jac_add_aff(ec::jac_mul(e, E), A3);
  ok[i] = (ec::jac_eq(l1, r1) && ec::jac_eq(l2, r2)) ? 1 : 0;
}
__global__ void __launch_bounds__(64) MPE_EC_OCC hash_commit_kernel(int B, ec::Enc enc, const uint32_t* __restrict__ P, const uint32_t* __restrict__ blind,
                                                         uint32_t* __restrict__ com) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  ec::u256_store(com + (size_t)i * 8, gg::commit_point(ec::aff_load(P + (size_t)i * 16), blind + (size_t)i * 8, enc));
}

}  // namespace mpe

extern "C" {

int mpe_pedersen_prove(mpe_ctx* ctx, int batch, const uint32_t* d_m, const uint32_t* d_r, const uint32_t* d_s1, const uint32_t* d_s2,
                       const mpe_pedersen_proof* out, void* stream) {
  if (!ctx || !d_m || !d_r || !d_s1 || !d_s2 || !out || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::pedersen_prove_kernel, batch, st, batch, ctx->enc, d_m, d_r, d_s1, d_s2, *out);
  return MPE_OK;
}
int mpe_pedersen_verify(mpe_ctx* ctx, int batch, const mpe_pedersen_proof* proof, uint8_t* d_ok, void* stream) {
  if (!ctx || !proof || !d_ok || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::pedersen_verify_kernel, batch, st, batch, ctx->enc, *proof, d_ok);
  return MPE_OK;
}
int mpe_heg_prove(mpe_ctx* ctx, int batch, const uint32_t* d_x, const uint32_t* d_r, const uint32_t* d_s1, const uint32_t* d_s2,
                  const mpe_heg_statement* statement, const mpe_heg_proof* out, void* stream) {
  if (!ctx || !d_x || !d_r || !d_s1 || !d_s2 || !statement || !out || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::heg_prove_kernel, batch, st, batch, ctx->enc, d_x, d_r, d_s1, d_s2, *statement, *out);
  return MPE_OK;
}
int mpe_heg_verify(mpe_ctx* ctx, int batch, const mpe_heg_statement* statement, const mpe_heg_proof* proof, uint8_t* d_ok, void* stream) {
  if (!ctx || !statement || !proof || !d_ok || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::heg_verify_kernel, batch, st, batch, ctx->enc, *statement, *proof, d_ok);
  return MPE_OK;
}
int mpe_hash_commit_point(mpe_ctx* ctx, int batch, const uint32_t* d_P, const uint32_t* d_blind, uint32_t* d_com, void* stream) {
  if (!ctx || !d_P || !d_blind || !d_com || batch < 0) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::hash_commit_kernel, batch, st, batch, ctx->enc, d_P, d_blind, d_com);
  return MPE_OK;
}

}  // extern "C"
