// Batched Paillier-2048 (kzen-paillier 0.4.2 call surface used by the reference:
// src/utilities/mta/mod.rs:22-24,68-75,133-145,165).  Included by mpe_lib.hip.
//
//   encrypt_with_chosen_randomness : c = (1 + m N) r^N mod N^2     1 exponentiation mod N^2 + 1 modmul
//   add                            : c1 c2 mod N^2                  1 modmul
//   mul                            : c^k mod N^2                    1 exponentiation mod N^2
//   decrypt (CRT)                  : m_p = L_p(c^(p-1) mod p^2) h_p mod p, same for q, recombine
//                                    2 exponentiations mod p^2 | q^2 + 4 modmul(2048) + two light kernels
// Every exponentiation modulo N^2 (public key only) or p^2 | q^2 (the key holder: modexp_nn) runs on the N-adic
// pair kernel of mpe_pairexp.h; the holder's x^N goes through (x^(q mod (p-1)) mod p)^p.  MPE_NO_PAIR / MPE_NO_CRT /
// MPE_NO_POWN switch back to the plain 4096- / 2048-bit kernels for A/B runs (tests check that every route agrees).
// All arithmetic runs on the GPU, including the per-key constants (h_p, h_q, CRT idempotents, pair constants),
// which the reference recomputes inside every decrypt call; per-key reuse is output-identical.
#pragma once
#include "mpe_internal.h"
#include "mpe_small.h"

struct mpe_paillier {
  int nkeys = 0;
  bool has_private = false;
  void* blob = nullptr;
  uint32_t* N = nullptr;      // [nk][64]
  uint32_t* NN = nullptr;     // [nk][128]
  mpe_modset* ms_nn = nullptr;  // 4096-bit, modulus k = N_k^2
  mpe_modset* ms_n = nullptr;   // 2048-bit, modulus k = N_k
  // private half-keys, j = 2k (p side) / 2k+1 (q side)
  uint32_t* pq32 = nullptr;   // [2nk][32]  p_k | q_k
  uint32_t* pq64 = nullptr;   // [2nk][64]  the same, zero-extended (moduli of ms_p)
  uint32_t* sq64 = nullptr;   // [2nk][64]  p^2 | q^2 (moduli of ms_pp)
  uint32_t* em1 = nullptr;    // [2nk][32]  p-1 | q-1
  uint32_t* em2 = nullptr;    // [2nk][32]  p-2 | q-2
  uint32_t* inv2 = nullptr;   // [2nk][32]  p^-1 | q^-1 mod 2^1024
  uint32_t* h64 = nullptr;    // [2nk][64]  h_p | h_q
  uint32_t* ab64 = nullptr;   // [2nk][64]  q (q^-1 mod p) | p (p^-1 mod q): the CRT idempotents mod N
  uint32_t* eq1 = nullptr;    // [2nk][32]  q mod (p-1) | p mod (q-1): x^N = (x^eq1 mod p)^p  (mod p^2)
  uint32_t* ecrt = nullptr;   // [2nk][64]  p(p-1)-1 | q(q-1)-1: the inverting exponent mod p^2 | q^2
  uint32_t* e128 = nullptr;   // [2nk][128] q^2 (q^-2 mod p^2) | p^2 (p^-2 mod q^2): the CRT idempotents mod N^2
  int32_t* swap_idx = nullptr;  // [2nk]    j ^ 1
  mpe_modset* ms_pp = nullptr;  // 2048-bit, moduli p^2 | q^2
  mpe_modset* ms_p = nullptr;   // 2048-bit, moduli p | q
  mpe_pairset* ps_nn = nullptr; // N-adic pair arithmetic modulo N_k^2   (mpe_pairexp.h)
  mpe_pairset* ps_pp = nullptr; // p-adic pair arithmetic modulo p^2 | q^2 (private key sets)
};

namespace mpe {

__device__ __forceinline__ int key_of(const int32_t* key_idx, int nkeys, int i) {
  return key_idx ? key_idx[i] : (nkeys == 1 ? 0 : i);
}

__global__ void pk_square_kernel(int nk, const uint32_t* __restrict__ N, uint32_t* __restrict__ NN) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nk) return;
  uint32_t n[64], r[128];
  sm::copy(n, N + (size_t)k * 64, 64);
  sm::mul(r, n, 64, n, 64);
  sm::copy(NN + (size_t)k * 128, r, 128);
}

// one lane per half-key j
__global__ void sk_setup_a_kernel(int nk, const uint32_t* __restrict__ p, const uint32_t* __restrict__ q,
                                  uint32_t* __restrict__ N, uint32_t* __restrict__ pq32, uint32_t* __restrict__ pq64,
                                  uint32_t* __restrict__ sq64, uint32_t* __restrict__ em1, uint32_t* __restrict__ em2,
                                  uint32_t* __restrict__ inv2, uint32_t* __restrict__ ecrt, uint32_t* __restrict__ eq1,
                                  int32_t* __restrict__ swap_idx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * nk) return;
  const int k = j >> 1;
  uint32_t x[32], y[32], r[64], t1[32], t2[32], one[1] = {1}, two[1] = {2};
  sm::copy(x, ((j & 1) ? q : p) + (size_t)k * 32, 32);
  sm::copy(y, ((j & 1) ? p : q) + (size_t)k * 32, 32);
  sm::copy(pq32 + (size_t)j * 32, x, 32);
  sm::copy(pq64 + (size_t)j * 64, x, 32);
  sm::zero(pq64 + (size_t)j * 64 + 32, 32);
  sm::mul(r, x, 32, x, 32);
  sm::copy(sq64 + (size_t)j * 64, r, 64);
  sm::sub(t1, 32, x, 32, one, 1);
  sm::copy(em1 + (size_t)j * 32, t1, 32);
  {                                                // other prime mod (own - 1): at most two subtractions
    uint32_t e[32];
    sm::copy(e, y, 32);
    for (int it = 0; it < 3; ++it)
      if (sm::cmp(e, 32, t1, 32) >= 0) sm::sub(e, 32, e, 32, t1, 32);
    sm::copy(eq1 + (size_t)j * 32, e, 32);
  }
  sm::mul(r, x, 32, t1, 32);                       // phi(x^2) = x (x - 1)
  sm::sub(r, 64, r, 64, one, 1);
  sm::copy(ecrt + (size_t)j * 64, r, 64);
  sm::sub(t1, 32, x, 32, two, 1);
  sm::copy(em2 + (size_t)j * 32, t1, 32);
  uint32_t iv[32];
  sm::inv2adic(iv, x, 32, t1, t2);
  sm::copy(inv2 + (size_t)j * 32, iv, 32);
  swap_idx[j] = j ^ 1;
  if ((j & 1) == 0) {
    sm::mul(r, x, 32, y, 32);
    sm::copy(N + (size_t)k * 64, r, 64);
  }
}

// inv_other[j] = (other prime)^-1 mod (own prime); h = own - inv_other; ab = other * inv_other
__global__ void sk_setup_b_kernel(int nk2, const uint32_t* __restrict__ pq32, const uint32_t* __restrict__ inv_other,
                                  uint32_t* __restrict__ h64, uint32_t* __restrict__ ab64) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nk2) return;
  uint32_t x[32], y[32], iv[32], r[64];
  sm::copy(x, pq32 + (size_t)j * 32, 32);
  sm::copy(y, pq32 + (size_t)(j ^ 1) * 32, 32);
  sm::copy(iv, inv_other + (size_t)j * 64, 32);
  sm::sub(r, 32, x, 32, iv, 32);
  sm::copy(h64 + (size_t)j * 64, r, 32);
  sm::zero(h64 + (size_t)j * 64 + 32, 32);
  sm::mul(r, y, 32, iv, 32);
  sm::copy(ab64 + (size_t)j * 64, r, 64);
}

// e128[j] = other^2 * inv_sq[j],  inv_sq[j] = (other^2)^-1 mod own^2
__global__ void sk_setup_c_kernel(int nk2, const uint32_t* __restrict__ sq64, const uint32_t* __restrict__ inv_sq,
                                  uint32_t* __restrict__ e128) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nk2) return;
  uint32_t a[64], b[64], r[128];
  sm::copy(a, sq64 + (size_t)(j ^ 1) * 64, 64);
  sm::copy(b, inv_sq + (size_t)j * 64, 64);
  sm::mul(r, a, 64, b, 64);
  sm::copy(e128 + (size_t)j * 128, r, 128);
}

// CRT plumbing for x^e mod N^2: half item j = 2i + side uses modulus 2 key(i) + side of ms_pp
__global__ void crt_half_kernel(int B2, Rows ksel, int32_t* __restrict__ half_of) {
  MPE_FOREGROUND();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= B2) return;
  half_of[j] = 2 * sel_index(ksel, j >> 1) + (j & 1);
}
// out = y[2i] + y[2i+1] mod N^2
__global__ void crt_add_kernel(int B, Rows ksel, const uint32_t* __restrict__ NN, const uint32_t* __restrict__ y,
                               uint32_t* __restrict__ out) {
  MPE_FOREGROUND();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t a[129], b[128];
  sm::copy(a, y + (size_t)(2 * i) * 128, 128);
  sm::copy(b, y + (size_t)(2 * i + 1) * 128, 128);
  a[128] = sm::add(a, 128, a, 128, b, 128);
  sm::copy(b, NN + (size_t)sel_index(ksel, i) * 128, 128);
  if (sm::cmp(a, 129, b, 128) >= 0) sm::sub(a, 129, a, 129, b, 128);
  sm::copy(out + (size_t)i * 128, a, 128);
}

// gm = 1 + m N   (128 words; m < N so gm < N^2)
__global__ void enc_gm_kernel(int B, int nkeys, const uint32_t* __restrict__ m, const int32_t* __restrict__ key_idx,
                              const uint32_t* __restrict__ N, uint32_t* __restrict__ gm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t a[64], n[64], r[128], one[1] = {1};
  sm::copy(a, m + (size_t)i * 64, 64);
  sm::copy(n, N + (size_t)key_of(key_idx, nkeys, i) * 64, 64);
  sm::mul(r, a, 64, n, 64);
  sm::add(r, 128, r, 128, one, 1);
  sm::copy(gm + (size_t)i * 128, r, 128);
}

// decrypt plumbing: item j = 2i + half
__global__ void dec_index_kernel(int B2, int nkeys, const int32_t* __restrict__ key_idx, int32_t* __restrict__ item_of,
                                 int32_t* __restrict__ half_of, int32_t* __restrict__ keyj) {
  MPE_FOREGROUND();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= B2) return;
  const int i = j >> 1, k = key_of(key_idx, nkeys, i);
  item_of[j] = i;
  half_of[j] = 2 * k + (j & 1);
  keyj[j] = k;
}
// t = (u - 1) / prime  (exact) = ((u - 1) mod 2^1024) * prime^-1 mod 2^1024     [L function]
__global__ void dec_lfunc_kernel(int B2, const uint32_t* __restrict__ u, const int32_t* __restrict__ half_of,
                                 const uint32_t* __restrict__ inv2, uint32_t* __restrict__ t) {
  MPE_FOREGROUND();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= B2) return;
  uint32_t a[32], iv[32], r[32], one[1] = {1};
  sm::copy(a, u + (size_t)j * 64, 32);
  sm::sub(a, 32, a, 32, one, 1);
  sm::copy(iv, inv2 + (size_t)half_of[j] * 32, 32);
  sm::mullo(r, a, iv, 32);
  sm::copy(t + (size_t)j * 32, r, 32);
}
// m = y[2i] + y[2i+1] mod N
__global__ void dec_combine_kernel(int B, int nkeys, const uint32_t* __restrict__ y, const int32_t* __restrict__ key_idx,
                                   const uint32_t* __restrict__ N, uint32_t* __restrict__ m) {
  MPE_FOREGROUND();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t a[65], b[64], n[64];
  sm::copy(a, y + (size_t)(2 * i) * 64, 64);
  sm::copy(b, y + (size_t)(2 * i + 1) * 64, 64);
  sm::copy(n, N + (size_t)key_of(key_idx, nkeys, i) * 64, 64);
  a[64] = sm::add(a, 64, a, 64, b, 64);
  if (sm::cmp(a, 65, n, 64) >= 0) sm::sub(a, 65, a, 65, n, 64);
  sm::copy(m + (size_t)i * 64, a, 64);
}

#define MPE_LAUNCH_1D(kernel, n, st, ...)                                                        \
  do {                                                                                            \
    if ((n) > 0) {                                                                                \
      hipLaunchKernelGGL(kernel, dim3(blocks_for((n), 64)), dim3(64), 0, st, __VA_ARGS__);        \
      hipError_t e_ = hipGetLastError();                                                          \
      if (e_ != hipSuccess) { mpe_set_error(#kernel, e_); return MPE_E_HIP; }                     \
    }                                                                                             \
  } while (0)
#define MPE_TRY(expr) do { int rc_ = (expr); if (rc_ != MPE_OK) return rc_; } while (0)

static void paillier_free(mpe_paillier* pk) {
  if (!pk) return;
  if (pk->ms_nn) mpe_modset_destroy(pk->ms_nn);
  if (pk->ms_n) mpe_modset_destroy(pk->ms_n);
  if (pk->ms_pp) mpe_modset_destroy(pk->ms_pp);
  if (pk->ms_p) mpe_modset_destroy(pk->ms_p);
  pairset_free(pk->ps_nn);
  pairset_free(pk->ps_pp);
  if (pk->blob) (void)hipFree(pk->blob);
  delete pk;
}

static int paillier_create(mpe_ctx* ctx, int nk, const uint32_t* d_N, const uint32_t* d_p, const uint32_t* d_q,
                           mpe_paillier** out, hipStream_t st) {
  mpe_paillier* pk = new (std::nothrow) mpe_paillier();
  if (!pk) return MPE_E_NOMEM;
  pk->nkeys = nk;
  pk->has_private = d_p != nullptr;
  const size_t nk2 = 2 * (size_t)nk;
  size_t words = (size_t)nk * (64 + 128);
  if (pk->has_private) words += nk2 * (32 + 64 + 64 + 32 + 32 + 32 + 64 + 64 + 64 + 128 + 32 + 1);
  hipError_t e = hipMalloc(&pk->blob, words * 4);
  if (e != hipSuccess) { delete pk; mpe_set_error("hipMalloc(paillier keys)", e); return MPE_E_NOMEM; }
  uint32_t* w = (uint32_t*)pk->blob;
  pk->N = w; w += (size_t)nk * 64;
  pk->NN = w; w += (size_t)nk * 128;
  int rc = MPE_OK;
  auto fail = [&](int code) { paillier_free(pk); return code; };
  if (pk->has_private) {
    pk->pq32 = w; w += nk2 * 32;
    pk->pq64 = w; w += nk2 * 64;
    pk->sq64 = w; w += nk2 * 64;
    pk->em1 = w; w += nk2 * 32;
    pk->em2 = w; w += nk2 * 32;
    pk->inv2 = w; w += nk2 * 32;
    pk->h64 = w; w += nk2 * 64;
    pk->ab64 = w; w += nk2 * 64;
    pk->ecrt = w; w += nk2 * 64;
    pk->eq1 = w; w += nk2 * 32;
    pk->e128 = w; w += nk2 * 128;
    pk->swap_idx = (int32_t*)w; w += nk2;
    hipLaunchKernelGGL(sk_setup_a_kernel, dim3(blocks_for((int)nk2, 64)), dim3(64), 0, st, nk, d_p, d_q, pk->N, pk->pq32,
                       pk->pq64, pk->sq64, pk->em1, pk->em2, pk->inv2, pk->ecrt, pk->eq1, pk->swap_idx);
    if ((rc = modset_create_dev(ctx, 2048, (int)nk2, pk->sq64, &pk->ms_pp, st)) != MPE_OK) return fail(rc);
    if ((rc = modset_create_dev(ctx, 2048, (int)nk2, pk->pq64, &pk->ms_p, st)) != MPE_OK) return fail(rc);
    // (other prime)^(own-2) mod own = (other prime)^-1 mod own   (Fermat; own is prime)
    if ((rc = ws_reserve(ctx, nk2 * 128 * 4 + 4096, st)) != MPE_OK) return fail(rc);
    uint32_t* inv_other = ws_array<uint32_t>(ctx, nk2 * 64);
    uint32_t* inv_sq = ws_array<uint32_t>(ctx, nk2 * 64);
    rc = launch_modexp(ctx, pk->ms_p, (int)nk2, rows(nullptr, 1), rows(pk->pq32, 32, pk->swap_idx, 32), no_rows(),
                       rows(pk->em2, 32), 32, inv_other, st);
    if (rc != MPE_OK) return fail(rc);
    hipLaunchKernelGGL(sk_setup_b_kernel, dim3(blocks_for((int)nk2, 64)), dim3(64), 0, st, (int)nk2, pk->pq32, inv_other,
                       pk->h64, pk->ab64);
    // (other^2)^(phi(own^2)-1) mod own^2 = (other^2)^-1 mod own^2
    rc = launch_modexp(ctx, pk->ms_pp, (int)nk2, rows(nullptr, 1), rows(pk->sq64, 64, pk->swap_idx, 64), no_rows(),
                       rows(pk->ecrt, 64), 64, inv_sq, st);
    if (rc != MPE_OK) return fail(rc);
    hipLaunchKernelGGL(sk_setup_c_kernel, dim3(blocks_for((int)nk2, 64)), dim3(64), 0, st, (int)nk2, pk->sq64, inv_sq, pk->e128);
  } else {
    e = hipMemcpyAsync(pk->N, d_N, (size_t)nk * 64 * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { mpe_set_error("hipMemcpyAsync(N)", e); return fail(MPE_E_HIP); }
  }
  hipLaunchKernelGGL(pk_square_kernel, dim3(blocks_for(nk, 64)), dim3(64), 0, st, nk, pk->N, pk->NN);
  if ((rc = modset_create_dev(ctx, 4096, nk, pk->NN, &pk->ms_nn, st)) != MPE_OK) return fail(rc);
  if ((rc = modset_create_dev(ctx, 2048, nk, pk->N, &pk->ms_n, st)) != MPE_OK) return fail(rc);
  if ((rc = pairset_create(2048, nk, pk->N, &pk->ps_nn, st)) != MPE_OK) return fail(rc);
  if (pk->has_private && (rc = pairset_create(1024, (int)nk2, pk->pq32, &pk->ps_pp, st)) != MPE_OK) return fail(rc);
  e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("paillier key set-up", e); return fail(MPE_E_HIP); }
  *out = pk;
  return MPE_OK;
}

static Rows key_selector(const mpe_paillier* pk, const int32_t* d_key_idx) {
  return Rows{nullptr, d_key_idx, (d_key_idx == nullptr && pk->nkeys != 1) ? 1 : 0, 0};
}
static Rows key_rows(const mpe_paillier* pk, const uint32_t* table, int stride, const int32_t* d_key_idx, int words = 0) {
  // per-item row of a per-key table
  if (d_key_idx) return Rows{table, d_key_idx, stride, words};
  return Rows{table, nullptr, pk->nkeys == 1 ? 0 : stride, words};
}

// words of workspace per item that modexp_nn takes when it goes through the CRT
constexpr size_t CRT_WS_WORDS = 2 + 2 * 64 + 2 * 128 + 192;

// base^exps mod N_k^2.  holder = the caller is the owner of the key (it may use p and q): the two halves
// mod p^2 | q^2 on the 2048-bit engine, then  x = x_p E_p + x_q E_q mod N^2.  Same residue as the direct form.
// pow_n: the exponent is N itself.  Then x^N = (x^q)^p and x^q = a (mod p) with a = x^(q mod (p-1)) mod p gives
// x^N = a^p (mod p^2): a 1024-bit exponentiation modulo p (one pass per squaring) and one modulo p^2 (two passes)
// instead of a 2048-bit one modulo p^2 — a quarter less work, the same residue.
// scratch: 2 B x (1 + 64 + 128) words of the caller's own (a launch that outlives the workspace reservation it was queued under: the PDL
// proofs' beta^N started rounds ahead, mpe_gg20.h round2); nullptr: the context workspace
static inline size_t modexp_nn_scratch_words(size_t B) { return 2 * B * (1 + 64 + 128) + 256; }
static int modexp_nn(mpe_ctx* ctx, const mpe_paillier* pk, int B, Rows ksel, Rows base, Rows exps, int ew, bool holder,
                     uint32_t* out, hipStream_t st, bool pow_n = false, uint32_t* scratch = nullptr, int phase = 0) {
  // phase (the key holder's x^N with scratch of the caller's): 1 = the first of the two ladders only, 2 = the rest, 0 = everything
  if (!(holder && pk->has_private && ctx->use_crt)) {
    // the exponent rows ARE the public-key table: x^N for a public N (r^N, s^N) — the one case that may run on sliding windows
    if (ctx->use_pair) return launch_pair_modexp(ctx, pk->ps_nn, B, ksel, base, exps, ew, no_rows(), no_rows(), 0, out, st, 0, exps.p == pk->N);
    return launch_modexp(ctx, pk->ms_nn, B, ksel, base, no_rows(), exps, ew, out, st);
  }
  const int B2 = 2 * B;
  int32_t* half_of = scratch ? (int32_t*)scratch : ws_array<int32_t>(ctx, B2);
  uint32_t* u = scratch ? scratch + (((size_t)B2 + 63) & ~(size_t)63) : ws_array<uint32_t>(ctx, (size_t)B2 * 64);
  uint32_t* y = scratch ? u + (size_t)B2 * 64 : ws_array<uint32_t>(ctx, (size_t)B2 * 128);
  if (!half_of || !u || !y) { mpe_set_error_msg("workspace under-reserved (modexp_nn)"); return MPE_E_NOMEM; }
  if (phase != 2) MPE_LAUNCH_1D(crt_half_kernel, B2, st, B2, ksel, half_of);
  const int bw = base.words ? base.words : 128;
  Rows lo{base.p, base.idx, base.stride, bw < 64 ? bw : 64, 1};
  Rows hi = bw > 64 ? Rows{base.p + 64, base.idx, base.stride, bw - 64, 1} : no_rows();
  Rows ex{exps.p, exps.idx, exps.stride, exps.words, 1};
  if (ctx->use_pair && pow_n && ctx->use_pown) {
    const Rows hsel{nullptr, half_of, 0, 0};
    if (phase != 2)
      MPE_TRY(launch_pair_modexp(ctx, pk->ps_pp, B2, hsel, Rows{base.p, base.idx, base.stride, bw, 1}, rows(pk->eq1, 32, half_of), 32,
                                 no_rows(), no_rows(), 0, y, st, 1));                      // a = x^(q mod (p-1)) mod p
    if (phase == 1) return MPE_OK;
    MPE_TRY(launch_pair_modexp(ctx, pk->ps_pp, B2, hsel, rows(y, 64, nullptr, 32), rows(pk->pq32, 32, half_of), 32,
                               no_rows(), no_rows(), 0, u, st));                           // a^p mod p^2
  } else if (ctx->use_pair) {
    MPE_TRY(launch_pair_modexp(ctx, pk->ps_pp, B2, Rows{nullptr, half_of, 0, 0}, Rows{base.p, base.idx, base.stride, bw, 1}, ex, ew,
                               no_rows(), no_rows(), 0, u, st));
  } else {
    MPE_TRY(launch_modexp(ctx, pk->ms_pp, B2, Rows{nullptr, half_of, 0, 0}, lo, hi, ex, ew, u, st));
  }
  MPE_TRY(launch_modmul(ctx, pk->ms_nn, B2, Rows{nullptr, ksel.idx, ksel.stride, 0, 1}, rows(u, 64, nullptr, 64),
                        rows(pk->e128, 128, half_of), y, st));
  MPE_LAUNCH_1D(crt_add_kernel, B, st, B, ksel, pk->NN, y, out);
  return MPE_OK;
}

// base^exps * base2^exps2 mod N_k^2 on one ladder, for a party that does NOT own the key (the verifiers, MessageB)
static int modexp_nn2(mpe_ctx* ctx, const mpe_paillier* pk, int B, Rows ksel, Rows base, Rows exps, int ew, Rows base2,
                      Rows exps2, int ew2, uint32_t* out, hipStream_t st) {
  if (ctx->use_pair) return launch_pair_modexp(ctx, pk->ps_nn, B, ksel, base, exps, ew, base2, exps2, ew2, out, st, 0, exps.p == pk->N);
  return launch_modexp2(ctx, pk->ms_nn, B, ksel, base, exps, ew, base2, exps2, ew2, out, st);
}

// x^e mod N for the HOLDER of N = p q (the provers' s = r^e beta mod N: range_proofs.rs:84, zk_pdl_with_slack/mod.rs:113): x^e mod p and
// x^e mod q in half mode on the 1024-bit pair engine (one pass per multiplication on half the limbs: a quarter of the multiply-adds of the
// 2048-bit ladder, and 36 instead of 72 steps of latency per multiplication for a small batch), recombined with the CRT idempotents as in
// paillier_decrypt.  Same residue.  Scratch: MODEXP_N_HOLDER_WS_WORDS words per item from the context workspace.
constexpr size_t MODEXP_N_HOLDER_WS_WORDS = 2 * (3 + 64 + 64) + 64;
static int modexp_n_holder(mpe_ctx* ctx, const mpe_paillier* pk, int B, const int32_t* key_idx, Rows base, Rows exps, int ew, uint32_t* out,
                           hipStream_t st) {
  const int B2 = 2 * B;
  int32_t* item_of = ws_array<int32_t>(ctx, B2);
  int32_t* half_of = ws_array<int32_t>(ctx, B2);
  int32_t* keyj = ws_array<int32_t>(ctx, B2);
  uint32_t* u = ws_array<uint32_t>(ctx, (size_t)B2 * 64);
  uint32_t* y = ws_array<uint32_t>(ctx, (size_t)B2 * 64);
  if (!item_of || !half_of || !keyj || !u || !y) { mpe_set_error_msg("workspace under-reserved (modexp_n_holder)"); return MPE_E_NOMEM; }
  MPE_LAUNCH_1D(dec_index_kernel, B2, st, B2, pk->nkeys, key_idx, item_of, half_of, keyj);
  MPE_TRY(launch_pair_modexp(ctx, pk->ps_pp, B2, Rows{nullptr, half_of, 0, 0}, Rows{base.p, base.idx, base.stride, base.words ? base.words : 64, 1},
                             Rows{exps.p, exps.idx, exps.stride, exps.words, 1}, ew, no_rows(), no_rows(), 0, u, st, 1));
  MPE_TRY(launch_modmul(ctx, pk->ms_n, B2, Rows{nullptr, keyj, 0, 0}, rows(u, 64, nullptr, 32), rows(pk->ab64, 64, half_of), y, st));
  MPE_LAUNCH_1D(dec_combine_kernel, B, st, B, pk->nkeys, y, key_idx, pk->N, out);
  return MPE_OK;
}

// workspace needs of the composites (a caller that runs several of them concurrently reserves the sum once)
static inline size_t ws_need_encrypt(int B) { return (size_t)B * (256 + CRT_WS_WORDS) * 4 + 8192; }
static inline size_t ws_need_mul_add_enc(int B) { return (size_t)B * 128 * 4 * 3 + 8192; }

// rn_pre: r^N mod N^2 already computed by the caller (a small-batch round merges every x^N of its key holders into one launch)
static int paillier_encrypt(mpe_ctx* ctx, const mpe_paillier* pk, int B, const int32_t* key_idx, const uint32_t* d_m,
                            const uint32_t* d_r, uint32_t* d_c, bool holder, hipStream_t st, const uint32_t* rn_pre = nullptr) {
  MPE_TRY(ws_reserve(ctx, ws_need_encrypt(B), st));
  uint32_t* x = ws_array<uint32_t>(ctx, (size_t)B * 128);
  uint32_t* gm = ws_array<uint32_t>(ctx, (size_t)B * 128);
  // r^N mod N^2
  if (rn_pre) x = const_cast<uint32_t*>(rn_pre);
  else
    MPE_TRY(modexp_nn(ctx, pk, B, key_selector(pk, key_idx), rows(d_r, 64, nullptr, 64), key_rows(pk, pk->N, 64, key_idx), 64,
                      holder, x, st, true));
  MPE_LAUNCH_1D(enc_gm_kernel, B, st, B, pk->nkeys, d_m, key_idx, pk->N, gm);
  return launch_modmul(ctx, pk->ms_nn, B, key_selector(pk, key_idx), rows(x, 128), rows(gm, 128), d_c, st);
}

// MessageB's ciphertext in one go: out = c_a^k * Enc(m; r) = c_a^k (1 + m N) r^N mod N^2   (mta/mod.rs:133-145:
// Paillier::encrypt_with_chosen_randomness, Paillier::mul, Paillier::add).  The peer computes this under a key
// it does not own; r^N and c_a^k share one ladder.
// x_pre: c_a^k r^N mod N^2 already computed by the caller (Round 1 merges these ladders with those of its verifications)
static int paillier_mul_add_enc(mpe_ctx* ctx, const mpe_paillier* pk, int B, const int32_t* key_idx, Rows c_a, Rows k, int kw,
                                const uint32_t* d_m, const uint32_t* d_r, uint32_t* d_out, hipStream_t st, const uint32_t* x_pre = nullptr) {
  MPE_TRY(ws_reserve(ctx, ws_need_mul_add_enc(B), st));
  uint32_t* x = ws_array<uint32_t>(ctx, (size_t)B * 128);
  uint32_t* gm = ws_array<uint32_t>(ctx, (size_t)B * 128);
  const Rows ksel = key_selector(pk, key_idx), Nrow = key_rows(pk, pk->N, 64, key_idx);
  MPE_LAUNCH_1D(enc_gm_kernel, B, st, B, pk->nkeys, d_m, key_idx, pk->N, gm);
  if (x_pre) {
    x = const_cast<uint32_t*>(x_pre);
  } else if (ctx->use_multiexp) {
    MPE_TRY(modexp_nn2(ctx, pk, B, ksel, rows(d_r, 64, nullptr, 64), Nrow, 64, c_a, k, kw, x, st));
  } else {
    uint32_t* y = ws_array<uint32_t>(ctx, (size_t)B * 128);
    MPE_TRY(modexp_nn(ctx, pk, B, ksel, rows(d_r, 64, nullptr, 64), Nrow, 64, false, x, st));
    MPE_TRY(modexp_nn(ctx, pk, B, ksel, c_a, k, kw, false, y, st));
    MPE_TRY(launch_modmul(ctx, pk->ms_nn, B, ksel, rows(x, 128), rows(y, 128), x, st));
  }
  return launch_modmul(ctx, pk->ms_nn, B, ksel, rows(x, 128), rows(gm, 128), d_out, st);
}

// c: one 128-word ciphertext row per item (dense, or through c.idx / c.stride: the round pipeline decrypts in place
// out of a message slab)
static int paillier_decrypt(mpe_ctx* ctx, const mpe_paillier* pk, int B, const int32_t* key_idx, Rows c, uint32_t* d_m,
                            hipStream_t st) {
  const int B2 = 2 * B;
  MPE_TRY(ws_reserve(ctx, (size_t)B2 * (3 * 4 + (64 + 32 + 64 + 64) * 4) + 16384, st));
  int32_t* item_of = ws_array<int32_t>(ctx, B2);
  int32_t* half_of = ws_array<int32_t>(ctx, B2);
  int32_t* keyj = ws_array<int32_t>(ctx, B2);
  uint32_t* u = ws_array<uint32_t>(ctx, (size_t)B2 * 64);
  uint32_t* t = ws_array<uint32_t>(ctx, (size_t)B2 * 32);
  uint32_t* mh = ws_array<uint32_t>(ctx, (size_t)B2 * 64);
  uint32_t* y = ws_array<uint32_t>(ctx, (size_t)B2 * 64);
  MPE_LAUNCH_1D(dec_index_kernel, B2, st, B2, pk->nkeys, key_idx, item_of, half_of, keyj);
  // u = c^(p-1) mod p^2 | c^(q-1) mod q^2   (c is double-width for the 2048-bit engine); half-item j reads row j >> 1
  if (ctx->use_pair) {
    MPE_TRY(launch_pair_modexp(ctx, pk->ps_pp, B2, Rows{nullptr, half_of, 0, 0}, Rows{c.p, c.idx, c.stride, 128, 1},
                               rows(pk->em1, 32, half_of), 32, no_rows(), no_rows(), 0, u, st));
  } else {
    MPE_TRY(launch_modexp(ctx, pk->ms_pp, B2, Rows{nullptr, half_of, 0, 0}, Rows{c.p, c.idx, c.stride, 64, 1},
                          Rows{c.p + 64, c.idx, c.stride, 64, 1}, rows(pk->em1, 32, half_of), 32, u, st));
  }
  MPE_LAUNCH_1D(dec_lfunc_kernel, B2, st, B2, u, half_of, pk->inv2, t);
  // m_p = L_p(u) h_p mod p | m_q
  MPE_TRY(launch_modmul(ctx, pk->ms_p, B2, Rows{nullptr, half_of, 0, 0}, rows(t, 32, nullptr, 32),
                        rows(pk->h64, 64, half_of), mh, st));
  // CRT with idempotents: m = m_p A + m_q B mod N
  MPE_TRY(launch_modmul(ctx, pk->ms_n, B2, Rows{nullptr, keyj, 0, 0}, rows(mh, 64), rows(pk->ab64, 64, half_of), y, st));
  MPE_LAUNCH_1D(dec_combine_kernel, B, st, B, pk->nkeys, y, key_idx, pk->N, d_m);
  return MPE_OK;
}

}  // namespace mpe

extern "C" {

int mpe_paillier_create_public(mpe_ctx* ctx, int nkeys, const uint32_t* d_N, mpe_paillier** out, void* stream) {
  if (!ctx || !d_N || !out || nkeys <= 0) return MPE_E_ARG;
  return mpe::paillier_create(ctx, nkeys, d_N, nullptr, nullptr, out, (hipStream_t)stream);
}
int mpe_paillier_create_private(mpe_ctx* ctx, int nkeys, const uint32_t* d_p, const uint32_t* d_q, mpe_paillier** out,
                                void* stream) {
  if (!ctx || !d_p || !d_q || !out || nkeys <= 0) return MPE_E_ARG;
  return mpe::paillier_create(ctx, nkeys, nullptr, d_p, d_q, out, (hipStream_t)stream);
}
int mpe_paillier_destroy(mpe_paillier* pk) {
  if (!pk) return MPE_E_ARG;
  mpe::paillier_free(pk);
  return MPE_OK;
}
int mpe_paillier_nkeys(const mpe_paillier* pk) { return pk ? pk->nkeys : MPE_E_ARG; }
const uint32_t* mpe_paillier_n(const mpe_paillier* pk) { return pk ? pk->N : nullptr; }

int mpe_paillier_encrypt(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx, const uint32_t* d_m,
                         const uint32_t* d_r, uint32_t* d_c, void* stream) {
  if (!ctx || !pk || !d_m || !d_r || !d_c || batch < 0) return MPE_E_ARG;
  if (!d_key_idx && pk->nkeys != 1 && pk->nkeys < batch) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  // a key set created with its primes belongs to the caller: encrypt through the CRT
  return mpe::paillier_encrypt(ctx, pk, batch, d_key_idx, d_m, d_r, d_c, pk->has_private, (hipStream_t)stream);
}
int mpe_paillier_decrypt(mpe_ctx* ctx, const mpe_paillier* sk, int batch, const int32_t* d_key_idx, const uint32_t* d_c,
                         uint32_t* d_m, void* stream) {
  if (!ctx || !sk || !d_c || !d_m || batch < 0) return MPE_E_ARG;
  if (!sk->has_private) { mpe_set_error_msg("mpe_paillier_decrypt: key set has no private part"); return MPE_E_ARG; }
  if (!d_key_idx && sk->nkeys != 1 && sk->nkeys < batch) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  return mpe::paillier_decrypt(ctx, sk, batch, d_key_idx, mpe::rows(d_c, 128), d_m, (hipStream_t)stream);
}
int mpe_paillier_add(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx, const uint32_t* d_c1,
                     const uint32_t* d_c2, uint32_t* d_out, void* stream) {
  if (!ctx || !pk || !d_c1 || !d_c2 || !d_out || batch < 0) return MPE_E_ARG;
  if (!d_key_idx && pk->nkeys != 1 && pk->nkeys < batch) return MPE_E_ARG;
  return mpe::launch_modmul(ctx, pk->ms_nn, batch, mpe::key_selector(pk, d_key_idx), mpe::rows(d_c1, 128),
                            mpe::rows(d_c2, 128), d_out, (hipStream_t)stream);
}
int mpe_paillier_mul(mpe_ctx* ctx, const mpe_paillier* pk, int batch, const int32_t* d_key_idx, const uint32_t* d_c,
                     const uint32_t* d_k, int k_words, uint32_t* d_out, void* stream) {
  if (!ctx || !pk || !d_c || !d_k || !d_out || batch < 0 || k_words <= 0) return MPE_E_ARG;
  if (!d_key_idx && pk->nkeys != 1 && pk->nkeys < batch) return MPE_E_ARG;
  return mpe::modexp_nn(ctx, pk, batch, mpe::key_selector(pk, d_key_idx), mpe::rows(d_c, 128), mpe::rows(d_k, k_words), k_words,
                        false, d_out, (hipStream_t)stream);
}

}  // extern "C"
