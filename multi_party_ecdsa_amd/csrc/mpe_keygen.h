// Keygen VERIFICATION math, batched: what every party checks about every other party's first keygen
// messages and shares — src/protocols/multi_party_ecdsa/gg_2020/party_i.rs
//   :260-320  NiCorrectKeyProof::verify (11 x sigma^N mod N), CompositeDLogProof::verify x2, the hash commitment
//   :322-367  VerifiableSS::validate_share (Feldman), commitments[0] == y_i
//   :405-438  DLogProof::verify + get_commitments_to_xi
// and the PROVE side of the two zk-paillier proofs (party_i.rs:219-258: NiCorrectKeyProof::proof, CompositeDLogProof::prove) —
// modexp-shaped as well.  Prime generation stays on the host: sequential and data dependent, SURVEY.md §2 row 7.
// The two zk-paillier 0.4.3 proofs are un-vendored: their definitions are recalled.  Every key is its own
// modulus here (one modulus per item), so the moduli set is built per call.  Included by mpe_lib.hip.
#pragma once
#include "mpe_proofs.h"

namespace mpe {
namespace kg {

constexpr int CK_M2 = 11;

// zk-paillier compute_digest over BigInts as minimal big-endian bytes; one (key, i) per lane:
//   seed = H(N, salt, i);  acc = sum_j H(seed, j) << (256 j), j < bit_length(N)/256 + 1  ->  hi (words 64..71) | lo (words 0..63)
__global__ void __launch_bounds__(64) MPE_EC_OCC ck_rho_kernel(int B, ec::Enc enc, const uint32_t* __restrict__ N, uint32_t* __restrict__ lo, uint32_t* __restrict__ hi) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * CK_M2) return;
  const int b = g / CK_M2, i = g % CK_M2;
  const uint32_t* n = N + (size_t)b * 64;
  ec::Sha256 s; ec::sha_init(s);
  ec::sha_bigint(s, n, 64, enc);
  const uint32_t salt[1] = {enc.ck_salt};                       // SALT_STRING = [75, 90, 101, 110] as a BigInt = 0x4B5A656E
  ec::sha_bigint(s, salt, 1, enc);
  const uint32_t iw[1] = {(uint32_t)i};
  ec::sha_bigint(s, iw, 1, enc);                                // i = 0: BigInt::to_bytes(0), enc.zero_bytes
  const ec::U256 seed = ec::sha_final(s);
  int top = 63;
  while (top > 0 && n[top] == 0) --top;
  const int bits = top * 32 + (32 - __clz(n[top] | 1u));
  const int msklen = bits / 256 + 1;
  uint32_t acc[72];
  for (int w = 0; w < 72; ++w) acc[w] = 0;
  for (int j = 0; j < msklen && j < 9; ++j) {
    ec::Sha256 t; ec::sha_init(t);
    ec::sha_bigint(t, seed.w, 8, enc);
    const uint32_t jw[1] = {(uint32_t)j};
    ec::sha_bigint(t, jw, 1, enc);
    const ec::U256 d = ec::sha_final(t);
    const int slot = enc.ck_mask_order ? (msklen < 9 ? msklen : 9) - 1 - j : j;      // block j at bit 256 j, or block 0 on top
    for (int w = 0; w < 8; ++w) acc[slot * 8 + w] = d.w[w];     // disjoint 256-bit slots: the sum is a concatenation
  }
  for (int w = 0; w < 64; ++w) lo[(size_t)g * 64 + w] = acc[w];
  for (int w = 0; w < 8; ++w) hi[(size_t)g * 8 + w] = acc[64 + w];
}
// key-level checks: N odd, > 1 and without a prime factor below 6370 (gcd(N, primorial) == 1, correct_key_ni.rs)
__global__ void __launch_bounds__(64) ck_small_factor_kernel(int B, const uint32_t* __restrict__ N, uint8_t* __restrict__ ok) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const uint32_t* n = N + (size_t)b * 64;
  bool good = (n[0] & 1u) != 0;
  for (uint32_t p = 3; p < 6370 && good; p += 2) {
    bool prime = true;
    for (uint32_t d = 3; d * d <= p; d += 2) if (p % d == 0) { prime = false; break; }
    if (!prime) continue;
    uint64_t r = 0;
    for (int w = 63; w >= 0; --w) r = ((r << 32) | n[w]) % p;
    if (r == 0) good = false;
  }
  ok[b] = good ? 1 : 0;
}
// rho = (hiT + lo_red) mod N and the comparison with sigma^N; ok[b] &= all 11
__global__ void ck_finish_kernel(int B, const uint32_t* __restrict__ N, const uint32_t* __restrict__ hiT, const uint32_t* __restrict__ lo_red,
                                 const uint32_t* __restrict__ sn, uint8_t* __restrict__ ok) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  bool good = ok[b] != 0;
  uint32_t n[64];
  sm::copy(n, N + (size_t)b * 64, 64);
  for (int i = 0; i < CK_M2 && good; ++i) {
    const size_t g = (size_t)b * CK_M2 + i;
    uint32_t a[65], c[64];
    sm::copy(a, hiT + g * 64, 64);
    sm::copy(c, lo_red + g * 64, 64);
    a[64] = sm::add(a, 64, a, 64, c, 64);
    if (sm::cmp(a, 65, n, 64) >= 0) sm::sub(a, 65, a, 65, n, 64);
    good = sm::cmp(a, 64, sn + g * 64, 64) == 0;
  }
  ok[b] = good ? 1 : 0;
}
__global__ void fill_words_kernel(int rows, int words, int bit, uint32_t* __restrict__ out) {       // rows of the constant 2^bit
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)rows * words) return;
  const int w = (int)(g % words);
  out[g] = (w == bit / 32) ? (1u << (bit % 32)) : 0u;
}
__global__ void iota_div_kernel(int n, int div, int32_t* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) out[g] = g / div;
}
// CompositeDLogProof: N >= 2^128 and odd
__global__ void cd_n_check_kernel(int B, const uint32_t* __restrict__ N, uint8_t* __restrict__ ok) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const uint32_t* n = N + (size_t)b * 64;
  uint32_t hi = 0;
  for (int w = 4; w < 64; ++w) hi |= n[w];
  ok[b] = ((n[0] & 1u) && hi) ? 1 : 0;
}
// Feldman: sum_k index^k C_k (Horner); share != null: ok = (share G == that), else the point itself
__global__ void __launch_bounds__(64) MPE_EC_OCC vss_kernel(int B, int t1, const uint32_t* __restrict__ commits, const uint32_t* __restrict__ share,
                                                 const int32_t* __restrict__ index, uint8_t* __restrict__ ok, uint32_t* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  ec::U256 idx = ec::u256_zero();
  idx.w[0] = (uint32_t)index[b];
  ec::Jac acc = ec::jac_inf();
  bool valid = true;
  for (int k = t1 - 1; k >= 0; --k) {
    const ec::Aff c = ec::aff_load(commits + ((size_t)b * t1 + k) * 16);
    valid = valid && ec::aff_valid(c);
    if (!ec::jac_is_inf(acc)) acc = ec::jac_mul(idx, ec::jac_to_aff(acc));
    acc = ec::jac_add_aff(acc, c);
  }
  if (share) ok[b] = (valid && ec::jac_eq(ec::jac_mul_gen(ec::sc_reduce(share + (size_t)b * 8, 8)), acc)) ? 1 : 0;
  if (out) ec::aff_store(out + (size_t)b * 16, ec::jac_to_aff(acc));
}

// rho = (hiT + lo_red) mod N, one (key, i) per lane
__global__ void ck_rho_sum_kernel(int n, const uint32_t* __restrict__ N, const uint32_t* __restrict__ hiT, const uint32_t* __restrict__ lo_red,
                                  uint32_t* __restrict__ rho) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  uint32_t a[65], c[64], m[64];
  sm::copy(m, N + (size_t)(g / CK_M2) * 64, 64);
  sm::copy(a, hiT + (size_t)g * 64, 64);
  sm::copy(c, lo_red + (size_t)g * 64, 64);
  a[64] = sm::add(a, 64, a, 64, c, 64);
  if (sm::cmp(a, 65, m, 64) >= 0) sm::sub(a, 65, a, 65, m, 64);
  sm::copy(rho + (size_t)g * 64, a, 64);
}
// y = r + e secret (plain integers: r 16 words, e 8, secret 64 -> 73 words)
__global__ void cd_y_kernel(int B, const uint32_t* __restrict__ r, const uint32_t* __restrict__ e, const uint32_t* __restrict__ secret,
                            uint32_t* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t t[73];
  sm::mul(t, e + (size_t)i * 8, 8, secret + (size_t)i * 64, 64);          // 72 words
  t[72] = 0;
  sm::add(t, 73, t, 73, r + (size_t)i * 16, 16);
  sm::copy(y + (size_t)i * 73, t, 73);
}
}  // namespace kg
}  // namespace mpe

namespace mpe {
namespace kg {
// the verdict of phase1_verify_com_phase3_verify_correct_key_verify_dlog_phase2_distribute for one prover (party_i.rs:277-303), in the
// reference's order of conjunction: commitment, NiCorrectKeyProof, bit lengths of e.n and dlog_statement.N, both CompositeDLogProofs
__global__ void r1_verdict_kernel(int B, int n, const uint32_t* __restrict__ com_want, const uint32_t* __restrict__ com_got, const uint8_t* __restrict__ ck,
                                  const uint32_t* __restrict__ N, const uint32_t* __restrict__ Nt, const uint8_t* __restrict__ cd1,
                                  const uint8_t* __restrict__ cd2, uint8_t* __restrict__ ok, uint32_t* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  bool v = sm::cmp(com_want + (size_t)i * 8, 8, com_got + (size_t)i * 8, 8) == 0 && ck[i];
  // PAILLIER_MIN_BIT_LENGTH = 2047 <= bit_length <= PAILLIER_MAX_BIT_LENGTH = 2048 (party_i.rs:49-50,287-290); a 64-word row cannot exceed 2048
  v = v && (N[(size_t)i * 64 + 63] >> 30) != 0 && (Nt[(size_t)i * 64 + 63] >> 30) != 0;
  v = v && cd1[i] && cd2[i];
  ok[i] = v ? 1 : 0;
  if (!v && bad) atomicOr(bad + i / n, 1u << (i % n));
}
// phase2_verify_vss_construct_keypair_phase3_pok_dlog (party_i.rs:337-343): validate_share && commitments[0] == y_vec[i]
__global__ void r2_verdict_kernel(int B, int n, int t1, const uint32_t* __restrict__ commits, const uint32_t* __restrict__ y, const uint8_t* __restrict__ share_ok,
                                  uint8_t* __restrict__ ok, uint32_t* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const bool v = share_ok[i] && sm::cmp(commits + (size_t)i * t1 * 16, 16, y + (size_t)i * 16, 16) == 0;
  ok[i] = v ? 1 : 0;
  if (!v && bad) atomicOr(bad + i / n, 1u << (i % n));
}
}  // namespace kg
}  // namespace mpe

extern "C" {

// NiCorrectKeyProof::proof(dk, SALT_STRING) for every key of a private key set: sigma_i = rho_i^(N^-1 mod phi(N)) mod N, i < 11
int mpe_correct_key_prove(mpe_ctx* ctx, const mpe_paillier* sk, uint32_t* d_sigma, void* stream) {
  using namespace mpe;
  if (!ctx || !sk || !d_sigma) return MPE_E_ARG;
  if (!sk->has_private) { mpe_set_error_msg("mpe_correct_key_prove: key set has no private part"); return MPE_E_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int nk = sk->nkeys, M = kg::CK_M2, n = nk * M;
  MPE_TRY(ws_reserve(ctx, ((size_t)n * (64 * 4 + 8 + 1) + (size_t)nk * 64 * 5 + modinv_ws_words(sk->ms_n, nk)) * 4 + 65536, st));
  int32_t* key_of = ws_array<int32_t>(ctx, n);
  uint32_t *lo = ws_array<uint32_t>(ctx, (size_t)n * 64), *hi = ws_array<uint32_t>(ctx, (size_t)n * 8), *hiT = ws_array<uint32_t>(ctx, (size_t)n * 64),
           *lor = ws_array<uint32_t>(ctx, (size_t)n * 64), *rho = ws_array<uint32_t>(ctx, (size_t)n * 64), *two = ws_array<uint32_t>(ctx, (size_t)nk * 64),
           *T = ws_array<uint32_t>(ctx, (size_t)nk * 64), *phi = ws_array<uint32_t>(ctx, (size_t)nk * 64), *u = ws_array<uint32_t>(ctx, (size_t)nk * 64),
           *d = ws_array<uint32_t>(ctx, (size_t)nk * 64);
  uint8_t* ok = ws_array<uint8_t>(ctx, ((size_t)nk + 255) & ~(size_t)255);
  if (!key_of || !lo || !hi || !hiT || !lor || !rho || !two || !T || !phi || !u || !d || !ok) { mpe_set_error_msg("correct_key_prove: workspace"); return MPE_E_NOMEM; }
  const Rows ksel{nullptr, key_of, 0, 0};
  hipLaunchKernelGGL(kg::iota_div_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, n, M, key_of);
  hipLaunchKernelGGL(kg::ck_rho_kernel, dim3(blocks_for(n, 64)), dim3(64), 0, st, nk, ctx->enc, sk->N, lo, hi);
  // rho = acc mod N with acc = hi 2^2048 + lo, as the verifier derives it
  hipLaunchKernelGGL(kg::fill_words_kernel, dim3(blocks_for(nk * 64, 256)), dim3(256), 0, st, nk, 64, 1024, two);
  MPE_TRY(launch_modmul(ctx, sk->ms_n, nk, rows(nullptr, 1), rows(two, 64), rows(two, 64), T, st));
  MPE_TRY(launch_modmul(ctx, sk->ms_n, n, ksel, rows(hi, 8, nullptr, 8), rows(T, 64, key_of), hiT, st));
  MPE_TRY(launch_modmul(ctx, sk->ms_n, n, ksel, rows(lo, 64), rows(sk->ms_n->one_words, 0, nullptr, 1), lor, st));
  hipLaunchKernelGGL(kg::ck_rho_sum_kernel, dim3(blocks_for(n, 64)), dim3(64), 0, st, n, sk->N, hiT, lor, rho);
  // d = N^-1 mod phi(N) (the same derivation as mpe_paillier_open), sigma = rho^d mod N
  MPE_LAUNCH_1D(bl::open_phi_kernel, nk, st, nk, sk->N, sk->pq32, phi);
  MPE_TRY(launch_modinv(ctx, sk->ms_n, nk, rows(nullptr, 1), rows(phi, 64), u, ok, st));
  MPE_LAUNCH_1D(bl::open_d_kernel, nk, st, nk, sk->N, phi, u, ok, d);
  const int rc = launch_modexp(ctx, sk->ms_n, n, ksel, rows(rho, 64), no_rows(), rows(d, 64, key_of), 64, d_sigma, st);
  (void)hipMemsetAsync(d, 0, (size_t)nk * 64 * 4, st);
  (void)hipMemsetAsync(phi, 0, (size_t)nk * 64 * 4, st);
  (void)hipMemsetAsync(u, 0, (size_t)nk * 64 * 4, st);
  return rc;
}

// CompositeDLogProof::prove(statement{N, g, ni}, secret) with the sampled r (< 2^512) as input: x = g^r mod N, e = H(x, g, N, ni),
// y = r + e secret.  d_secret [batch][64], d_r [batch][16]; outputs d_x [batch][64], d_y [batch][73].
int mpe_composite_dlog_prove(mpe_ctx* ctx, int batch, const uint32_t* d_N, const uint32_t* d_g, const uint32_t* d_ni, const uint32_t* d_secret,
                             const uint32_t* d_r, uint32_t* d_x, uint32_t* d_y, void* stream) {
  using namespace mpe;
  if (!ctx || !d_N || !d_g || !d_ni || !d_secret || !d_r || !d_x || !d_y || batch < 0) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  mpe_modset* ms = nullptr;
  MPE_TRY(modset_create_dev(ctx, 2048, batch, d_N, &ms, st));
  int rc = ws_reserve(ctx, (size_t)batch * (64 + 8) * 4 + 65536, st);
  if (rc != MPE_OK) { mpe_modset_destroy(ms); return rc; }
  Seq q{ctx, st, batch};
  uint32_t* e = q.words(8);
  rc = launch_modexp(ctx, ms, batch, rows(nullptr, 1), rows(d_g, 64), no_rows(), rows(d_r, 16), 16, d_x, st);
  q.rc = rc;
  HashDesc d;
  d.n = 4;
  { const HashField canon[4] = {hf(rows(d_x, 64), 64), hf(rows(d_g, 64), 64), hf(rows(d_N, 64), 64), hf(rows(d_ni, 64), 64)};   // (x, g, N, ni)
    for (int k = 0; k < 4; ++k) d.f[k] = canon[ctx->enc.ord_cdlog[k] & 3]; }
  q.hash(d, e);
  if (q.rc == MPE_OK) hipLaunchKernelGGL(kg::cd_y_kernel, dim3(blocks_for(batch, 64)), dim3(64), 0, st, batch, d_r, e, d_secret, d_y);
  rc = q.finish("mpe_composite_dlog_prove");
  (void)hipStreamSynchronize(st);
  mpe_modset_destroy(ms);
  return rc;
}

int mpe_correct_key_verify(mpe_ctx* ctx, int batch, const uint32_t* d_N, const uint32_t* d_sigma, uint8_t* d_ok, void* stream) {
  if (!ctx || !d_N || !d_sigma || !d_ok || batch < 0) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  using namespace mpe;
  hipStream_t st = (hipStream_t)stream;
  const int M = kg::CK_M2, n = batch * M;
  mpe_modset* ms = nullptr;
  MPE_TRY(modset_create_dev(ctx, 2048, batch, d_N, &ms, st));
  int rc = ws_reserve(ctx, ((size_t)n * (64 * 5 + 8 + 1) + (size_t)batch * 64 * 3) * 4 + 65536, st);
  if (rc != MPE_OK) { mpe_modset_destroy(ms); return rc; }
  int32_t* key_of = ws_array<int32_t>(ctx, n);
  uint32_t *lo = ws_array<uint32_t>(ctx, (size_t)n * 64), *hi = ws_array<uint32_t>(ctx, (size_t)n * 8), *sn = ws_array<uint32_t>(ctx, (size_t)n * 64),
           *hiT = ws_array<uint32_t>(ctx, (size_t)n * 64), *lor = ws_array<uint32_t>(ctx, (size_t)n * 64), *two = ws_array<uint32_t>(ctx, (size_t)batch * 64),
           *T = ws_array<uint32_t>(ctx, (size_t)batch * 64);
  if (!key_of || !lo || !hi || !sn || !hiT || !lor || !two || !T) { mpe_modset_destroy(ms); mpe_set_error_msg("correct_key: workspace"); return MPE_E_NOMEM; }
  const Rows ksel{nullptr, key_of, 0, 0};
  hipLaunchKernelGGL(kg::iota_div_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, n, M, key_of);
  hipLaunchKernelGGL(kg::ck_small_factor_kernel, dim3(blocks_for(batch, 64)), dim3(64), 0, st, batch, d_N, d_ok);
  hipLaunchKernelGGL(kg::ck_rho_kernel, dim3(blocks_for(n, 64)), dim3(64), 0, st, batch, ctx->enc, d_N, lo, hi);
  // sigma_i^N mod N: the heavy part — 11 exponentiations (2048-bit modulus, 2048-bit exponent) per key
  rc = launch_modexp(ctx, ms, n, ksel, rows(d_sigma, 64), no_rows(), rows(d_N, 64, key_of), 64, sn, st);
  // rho = acc mod N with acc = hi 2^2048 + lo:  T = 2^2048 mod N = (2^1024)^2, rho = hi T + lo (mod N)
  hipLaunchKernelGGL(kg::fill_words_kernel, dim3(blocks_for(batch * 64, 256)), dim3(256), 0, st, batch, 64, 1024, two);
  if (rc == MPE_OK) rc = launch_modmul(ctx, ms, batch, rows(nullptr, 1), rows(two, 64), rows(two, 64), T, st);
  if (rc == MPE_OK) rc = launch_modmul(ctx, ms, n, ksel, rows(hi, 8, nullptr, 8), rows(T, 64, key_of), hiT, st);
  if (rc == MPE_OK) rc = launch_modmul(ctx, ms, n, ksel, rows(lo, 64), rows(ms->one_words, 0, nullptr, 1), lor, st);
  if (rc == MPE_OK) hipLaunchKernelGGL(kg::ck_finish_kernel, dim3(blocks_for(batch, 64)), dim3(64), 0, st, batch, d_N, hiT, lor, sn, d_ok);
  (void)hipStreamSynchronize(st);          // the per-call moduli set is released below
  mpe_modset_destroy(ms);
  if (rc != MPE_OK) return rc;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("mpe_correct_key_verify", e); return MPE_E_HIP; }
  return MPE_OK;
}

int mpe_composite_dlog_verify(mpe_ctx* ctx, int batch, const uint32_t* d_N, const uint32_t* d_g, const uint32_t* d_ni, const uint32_t* d_x,
                              const uint32_t* d_y, uint8_t* d_ok, void* stream) {
  if (!ctx || !d_N || !d_g || !d_ni || !d_x || !d_y || !d_ok || batch < 0) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  using namespace mpe;
  hipStream_t st = (hipStream_t)stream;
  mpe_modset* ms = nullptr;
  MPE_TRY(modset_create_dev(ctx, 2048, batch, d_N, &ms, st));
  int rc = ws_reserve(ctx, ((size_t)batch * (64 * 6 + 8) + modinv_ws_words(ms, batch) * 2) * 4 + 65536, st);
  if (rc != MPE_OK) { mpe_modset_destroy(ms); return rc; }
  Seq q{ctx, st, batch};
  const Rows sel = rows(nullptr, 1);                                    // modulus i for item i
  uint8_t* ok1 = q.flags();
  hipLaunchKernelGGL(kg::cd_n_check_kernel, dim3(blocks_for(batch, 64)), dim3(64), 0, st, batch, d_N, d_ok);
  // gcd(g, N) == gcd(ni, N) == 1  <=>  g ni is a unit modulo N: ONE inversion per item (its ok flag), not two — the moduli differ
  // per item, so the inversions cannot share a Montgomery batch and are the larger part of this call
  uint32_t* gn = q.modmul(ms, sel, rows(d_g, 64), rows(d_ni, 64));
  (void)q.modinv(ms, sel, rows(gn, 64), ok1);
  // e = H(x, g, N, ni);  x == g^y ni^e mod N
  uint32_t* e = q.words(8);
  HashDesc d;
  d.n = 4;
  { const HashField canon[4] = {hf(rows(d_x, 64), 64), hf(rows(d_g, 64), 64), hf(rows(d_N, 64), 64), hf(rows(d_ni, 64), 64)};   // (x, g, N, ni)
    for (int k = 0; k < 4; ++k) d.f[k] = canon[ctx->enc.ord_cdlog[k] & 3]; }
  q.hash(d, e);
  uint32_t* gy = q.modexp(ms, sel, rows(d_g, 64), rows(d_y, 73), 73);
  uint32_t* ne = q.modexp(ms, sel, rows(d_ni, 64), rows(e, 8), 8);
  uint32_t* pr = q.modmul(ms, sel, rows(gy, 64), rows(ne, 64));
  if (q.rc == MPE_OK) hipLaunchKernelGGL(and_flags_kernel, dim3(blocks_for(batch, 64)), dim3(64), 0, st, batch, d_ok, ok1, (const uint8_t*)nullptr, pr, rows(d_x, 64), 64);
  rc = q.finish("mpe_composite_dlog_verify");
  (void)hipStreamSynchronize(st);
  mpe_modset_destroy(ms);
  return rc;
}

int mpe_vss_validate_share(mpe_ctx* ctx, int batch, int t1, const uint32_t* d_commits, const uint32_t* d_share, const int32_t* d_index,
                           uint8_t* d_ok, void* stream) {
  if (!ctx || !d_commits || !d_share || !d_index || !d_ok || batch < 0 || t1 < 1 || t1 > 64) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::kg::vss_kernel, batch, st, batch, t1, d_commits, d_share, d_index, d_ok, (uint32_t*)nullptr);
  return MPE_OK;
}
int mpe_vss_point_commitment(mpe_ctx* ctx, int batch, int t1, const uint32_t* d_commits, const int32_t* d_index, uint32_t* d_out, void* stream) {
  if (!ctx || !d_commits || !d_index || !d_out || batch < 0 || t1 < 1 || t1 > 64) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  MPE_LAUNCH_1D(mpe::kg::vss_kernel, batch, st, batch, t1, d_commits, (const uint32_t*)nullptr, d_index, (uint8_t*)nullptr, d_out);
  return MPE_OK;
}


// `Keys::phase1_verify_com_phase3_verify_correct_key_verify_dlog_phase2_distribute` (party_i.rs:260-320) as the reference composes it
int mpe_keygen_verify_round1(mpe_ctx* ctx, int batch, int n_parties, const mpe_keygen_round1* in, uint8_t* d_ok, uint32_t* d_bad_actors, void* stream) {
  if (!ctx || !in || !d_ok || batch < 0 || n_parties < 1 || n_parties > 32 || batch % n_parties) return MPE_E_ARG;
  if (!in->y || !in->blind || !in->com || !in->N || !in->sigma || !in->Nt || !in->h1 || !in->h2 || !in->x_h1 || !in->y_h1 || !in->x_h2 || !in->y_h2) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  // the three verifiers use the context workspace one after another: their verdicts live in an allocation of this call
  uint8_t* flags = nullptr;
  if (hipMalloc((void**)&flags, (size_t)batch * (3 + 32) + 256) != hipSuccess) { mpe_set_error_msg("keygen round1: hipMalloc"); return MPE_E_NOMEM; }
  uint8_t *ck = flags, *cd1 = flags + batch, *cd2 = flags + 2 * (size_t)batch;
  uint32_t* com = (uint32_t*)(flags + (((size_t)3 * batch + 255) & ~(size_t)255));
  int rc = mpe_hash_commit_point(ctx, batch, in->y, in->blind, com, stream);                                    // party_i.rs:278-283
  if (rc == MPE_OK) rc = mpe_correct_key_verify(ctx, batch, in->N, in->sigma, ck, stream);                      // :284-287
  if (rc == MPE_OK) rc = mpe_composite_dlog_verify(ctx, batch, in->Nt, in->h1, in->h2, in->x_h1, in->y_h1, cd1, stream);   // :292-295
  if (rc == MPE_OK) rc = mpe_composite_dlog_verify(ctx, batch, in->Nt, in->h2, in->h1, in->x_h2, in->y_h2, cd2, stream);   // :296-299, g and ni swapped (:271-275)
  if (rc == MPE_OK) {
    if (d_bad_actors) (void)hipMemsetAsync(d_bad_actors, 0, (size_t)(batch / n_parties) * 4, st);
    hipLaunchKernelGGL(mpe::kg::r1_verdict_kernel, dim3(mpe::blocks_for(batch, 64)), dim3(64), 0, st, batch, n_parties, in->com, com, ck, in->N, in->Nt, cd1, cd2,
                       d_ok, d_bad_actors);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mpe_set_error("mpe_keygen_verify_round1", e); rc = MPE_E_HIP; }
  }
  (void)hipStreamSynchronize(st);
  (void)hipFree(flags);
  return rc;
}

// `Keys::phase2_verify_vss_construct_keypair_phase3_pok_dlog`, the verdict (party_i.rs:322-367)
int mpe_keygen_verify_round2(mpe_ctx* ctx, int batch, int n_parties, int t1, const uint32_t* d_commits, const uint32_t* d_share, const int32_t* d_index,
                             const uint32_t* d_y, uint8_t* d_ok, uint32_t* d_bad_actors, void* stream) {
  if (!ctx || !d_commits || !d_share || !d_index || !d_y || !d_ok || batch < 0 || n_parties < 1 || n_parties > 32 || batch % n_parties || t1 < 1 || t1 > 64) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  hipStream_t st = (hipStream_t)stream;
  uint8_t* share_ok = nullptr;
  if (hipMalloc((void**)&share_ok, (size_t)batch) != hipSuccess) { mpe_set_error_msg("keygen round2: hipMalloc"); return MPE_E_NOMEM; }
  int rc = mpe_vss_validate_share(ctx, batch, t1, d_commits, d_share, d_index, share_ok, stream);
  if (rc == MPE_OK) {
    if (d_bad_actors) (void)hipMemsetAsync(d_bad_actors, 0, (size_t)(batch / n_parties) * 4, st);
    hipLaunchKernelGGL(mpe::kg::r2_verdict_kernel, dim3(mpe::blocks_for(batch, 64)), dim3(64), 0, st, batch, n_parties, t1, d_commits, d_y, share_ok, d_ok, d_bad_actors);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mpe_set_error("mpe_keygen_verify_round2", e); rc = MPE_E_HIP; }
  }
  (void)hipStreamSynchronize(st);
  (void)hipFree(share_ok);
  return rc;
}

}  // extern "C"
