// Batched modular inversion (curv `BigInt::mod_inv`, src/utilities/mta/range_proofs.rs:122,135,339,351,363;
// src/utilities/zk_pdl_with_slack/mod.rs:192).  Included by mpe_lib.hip.
//
// The verifiers need (z^e)^-1 mod N~ and (c^e)^-1 mod N^2 for every item, but all items of a batch share a
// handful of moduli (one per key).  Montgomery's trick turns the B inversions of a launch into 5 Montgomery
// multiplications per item plus ONE real inversion per chunk of <= 64 same-modulus items:
//   1. bucket the items by modulus (device counting sort), cut the buckets into chunks;
//   2. up-sweep   (modexp engine, one lane group per chunk): prefix products P_j = x_0 ... x_j;
//   3. invert the chunk totals: one extended binary gcd per WAVE, limbs across the 64 lanes, carries resolved
//      with ballots — a few ms of latency per launch, all chunks concurrently;
//   4. down-sweep (modexp engine): x_j^-1 = T^-1 P_{j-1}, T^-1 <- T^-1 x_j.
// A chunk whose total is not invertible (some gcd(x_j, n) != 1) falls back to the lane-serial kernel for its
// items, so ok[] is exact per item.  Inputs must be reduced (x < n), which every call site guarantees.
#pragma once
#include "mpe_internal.h"
#include "mpe_small.h"

namespace mpe {

// ---------------------------------------------------------------------------------------------
// lane-serial binary extended gcd (fallback, and the path for per-item moduli)
// ---------------------------------------------------------------------------------------------
template <int K32>
__global__ void modinv_lane_kernel(int B, const uint32_t* __restrict__ mod_words, Rows mod_sel, Rows A,
                                   const uint8_t* __restrict__ only_if, uint32_t* __restrict__ out,
                                   uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  if (only_if && !only_if[i]) return;
  constexpr int W = K32 + 1;
  uint32_t u[W], v[W], x1[W], x2[W], m[W];
  const int mi = sel_index(mod_sel, i);
  const uint32_t* mp = mod_words + (size_t)mi * K32;
  const uint32_t* ap = row_of(A, i);
  const int aw = A.words ? A.words : K32;
  for (int j = 0; j < W; ++j) {
    m[j] = j < K32 ? mp[j] : 0;
    v[j] = m[j];
    u[j] = j < aw ? ap[j] : 0;
    x1[j] = j == 0 ? 1u : 0u;
    x2[j] = 0;
  }
  auto is_one = [&](const uint32_t* a) { uint32_t o = a[0] ^ 1u; for (int j = 1; j < W; ++j) o |= a[j]; return o == 0; };
  auto halve_mod = [&](uint32_t* x) {      // x <- x/2 mod m
    if (x[0] & 1u) sm::add(x, W, x, W, m, W);
    for (int j = 0; j < W - 1; ++j) x[j] = (x[j] >> 1) | (x[j + 1] << 31);
    x[W - 1] >>= 1;
  };
  auto shr1 = [&](uint32_t* x) {
    for (int j = 0; j < W - 1; ++j) x[j] = (x[j] >> 1) | (x[j + 1] << 31);
    x[W - 1] >>= 1;
  };
  bool good = !sm::is_zero(u, W);
  int guard = 4 * 32 * K32 + 8;
  while (good && !is_one(u) && !is_one(v) && guard-- > 0) {
    while (!(u[0] & 1u)) { shr1(u); halve_mod(x1); }
    while (!(v[0] & 1u)) { shr1(v); halve_mod(x2); }
    if (sm::cmp(u, W, v, W) >= 0) {
      sm::sub(u, W, u, W, v, W);
      if (sm::sub(x1, W, x1, W, x2, W)) sm::add(x1, W, x1, W, m, W);
      if (sm::is_zero(u, W)) good = false;            // gcd(a, m) = v != 1
    } else {
      sm::sub(v, W, v, W, u, W);
      if (sm::sub(x2, W, x2, W, x1, W)) sm::add(x2, W, x2, W, m, W);
    }
  }
  const uint32_t* res = is_one(u) ? x1 : x2;
  if (!is_one(u) && !is_one(v)) good = false;
  for (int j = 0; j < K32; ++j) out[(size_t)i * K32 + j] = good ? res[j] : 0u;
  ok[i] = good ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// wave-cooperative binary extended gcd: ONE inversion per wave, limb l of every operand in lane l
// (LT = uint32_t for 2048 bit, uint64_t for 4096 bit).  Every branch is wave-uniform.
// ---------------------------------------------------------------------------------------------
template <typename LT>
struct WaveInt {
  static constexpr int LB = sizeof(LT) * 8;
  // lane l <- lane l+1 (lane 63 <- 0)
  static __device__ __forceinline__ LT from_above(LT x, int lane) {
    LT y;
    if constexpr (sizeof(LT) == 8) {
      const uint32_t lo = __shfl_down((uint32_t)x, 1), hi = __shfl_down((uint32_t)(x >> 32), 1);
      y = ((uint64_t)hi << 32) | lo;
    } else {
      y = __shfl_down(x, 1);
    }
    return lane == 63 ? (LT)0 : y;
  }
  // carries / borrows that enter each lane, from per-lane generate and propagate flags
  static __device__ __forceinline__ uint64_t ripple(bool g, bool p, bool& out) {
    const uint64_t G = __ballot(g), P = __ballot(p);
    const uint64_t C = ((G << 1) + P) ^ P;
    out = ((G >> 63) & 1) | (((P >> 63) & 1) & ((C >> 63) & 1));
    return C;
  }
  static __device__ __forceinline__ bool add(LT& r, LT a, LT b, int lane) {
    const LT s = a + b;
    bool co;
    const uint64_t C = ripple(s < a, s == (LT)~(LT)0, co);
    r = s + (LT)((C >> lane) & 1);
    return co;
  }
  static __device__ __forceinline__ bool sub(LT& r, LT a, LT b, int lane) {
    const LT d = a - b;
    bool bo;
    const uint64_t C = ripple(a < b, d == 0, bo);
    r = d - (LT)((C >> lane) & 1);
    return bo;
  }
  static __device__ __forceinline__ bool ge(LT a, LT b) { return __ballot(a > b) >= __ballot(a < b); }
  static __device__ __forceinline__ void shr1(LT& x, bool top_in, int lane) {
    const LT up = from_above(x, lane);
    LT y = (x >> 1) | (up << (LB - 1));
    if (lane == 63 && top_in) y |= (LT)1 << (LB - 1);
    x = y;
  }
  static __device__ __forceinline__ bool is_small(LT x, unsigned v, int lane) {   // x == v (v = 0 or 1)
    return __ballot(x != (lane == 0 ? (LT)v : (LT)0)) == 0;
  }
};

template <typename LT>
__global__ void __launch_bounds__(64) modinv_wave_kernel(const int32_t* __restrict__ n_items, const uint32_t* __restrict__ mod_words,
                                                         const int32_t* __restrict__ mod_of, const uint32_t* __restrict__ a,
                                                         uint32_t* __restrict__ out, uint8_t* __restrict__ ok) {
  MPE_FOREGROUND();
  using WI = WaveInt<LT>;
  constexpr int WPL = sizeof(LT) / 4;            // interface words per lane
  constexpr int K32 = 64 * WPL;
  const int item = blockIdx.x;
  if (item >= *n_items) return;
  const int lane = threadIdx.x & 63;
  auto load = [&](const uint32_t* p) -> LT {
    if constexpr (WPL == 2) return (LT)p[2 * lane] | ((LT)p[2 * lane + 1] << 32);
    else return (LT)p[lane];
  };
  const LT m = load(mod_words + (size_t)mod_of[item] * K32);
  LT u = load(a + (size_t)item * K32), v = m, x1 = lane == 0 ? 1 : 0, x2 = 0;
  bool good = !WI::is_small(u, 0, lane);
  LT res = 0;
  auto halve = [&](LT& x) {                      // x <- x / 2 mod m
    bool c = false;
    const bool odd = __shfl((uint32_t)x, 0) & 1u;
    if (odd) c = WI::add(x, x, m, lane);
    WI::shr1(x, c, lane);
  };
  auto submod = [&](LT& x, LT y) {               // x <- x - y mod m   (x, y < m)
    if (WI::sub(x, x, y, lane)) (void)WI::add(x, x, m, lane);
  };
  int guard = 4 * 32 * K32 + 64;
  while (good) {
    if (WI::is_small(u, 1, lane)) { res = x1; break; }
    if (WI::is_small(v, 1, lane)) { res = x2; break; }
    if (guard-- <= 0) { good = false; break; }
    const bool u_odd = __shfl((uint32_t)u, 0) & 1u, v_odd = __shfl((uint32_t)v, 0) & 1u;
    if (!u_odd) { WI::shr1(u, false, lane); halve(x1); }
    else if (!v_odd) { WI::shr1(v, false, lane); halve(x2); }
    else if (WI::ge(u, v)) {
      (void)WI::sub(u, u, v, lane);
      submod(x1, x2);
      if (WI::is_small(u, 0, lane)) good = false;   // gcd = v != 1
    } else {
      (void)WI::sub(v, v, u, lane);
      submod(x2, x1);
    }
  }
  if (!good) res = 0;
  if constexpr (WPL == 2) { out[(size_t)item * K32 + 2 * lane] = (uint32_t)res; out[(size_t)item * K32 + 2 * lane + 1] = (uint32_t)(res >> 32); }
  else out[(size_t)item * K32 + lane] = (uint32_t)res;
  if (lane == 0) ok[item] = good ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// bucketing by modulus and chunk table
// ---------------------------------------------------------------------------------------------
struct InvPlan {
  int32_t* cnt;        // [nmod + 1]
  int32_t* rank;       // [B]
  int32_t* perm;       // [B]   position in the modulus-sorted order -> item
  int32_t* ch_start;   // [maxch]
  int32_t* ch_len;     // [maxch]
  int32_t* ch_mod;     // [maxch]
  int32_t* nch;        // [1]
};
__device__ __forceinline__ int mod_index(const Rows& sel, int i) { return sel_index(sel, i); }

// rank[i] = position of item i among the items of its modulus.  A launch has a handful of moduli, so the lanes of a
// wave are grouped by modulus with ballots and ONE lane per (wave, modulus) does the atomic: 64x fewer colliding
// atomics than one per item.
__global__ void inv_count_kernel(int B, Rows mod_sel, InvPlan p) {
  MPE_FOREGROUND();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < B;
  const int m = live ? mod_index(mod_sel, i) : -1;
  const int lane = threadIdx.x & 63;
  uint64_t todo = __ballot(live);
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const int lm = __shfl(m, leader);                    // the modulus this round serves
    const uint64_t same = __ballot(live && m == lm);
    int base = 0;
    if (lane == leader) base = atomicAdd(&p.cnt[lm], (int)__builtin_popcountll(same));
    base = __shfl(base, leader);
    if (live && m == lm) p.rank[i] = base + (int)__builtin_popcountll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
}
// one wave: bucket offsets (in place of the counts) and the chunk table.  Every lane takes a contiguous run of moduli, counts its items and
// chunks, the 64 partial sums are scanned, and the lane writes its run (a launch may carry tens of thousands of moduli — every
// session its own wallet — where one lane walking all of them took ~0.1 s per inversion call)
__global__ void __launch_bounds__(64) inv_plan_kernel(int nmod, int chunk, InvPlan p) {
  MPE_FOREGROUND();
  __shared__ int32_t s_items[64], s_chunks[64];
  const int lane = threadIdx.x, per = (nmod + 63) / 64, lo = lane * per, hi = lo + per < nmod ? lo + per : nmod;
  int items = 0, chunks = 0;
  for (int m = lo; m < hi; ++m) { const int c = p.cnt[m]; items += c; chunks += (c + chunk - 1) / chunk; }
  s_items[lane] = items; s_chunks[lane] = chunks;
  __syncthreads();
  int off = 0, nc = 0;
  for (int t = 0; t < lane; ++t) { off += s_items[t]; nc += s_chunks[t]; }
  for (int m = lo; m < hi; ++m) {
    const int c = p.cnt[m];
    p.cnt[m] = off;
    for (int s = 0; s < c; s += chunk) {
      p.ch_start[nc] = off + s;
      p.ch_len[nc] = c - s < chunk ? c - s : chunk;
      p.ch_mod[nc] = m;
      ++nc;
    }
    off += c;
  }
  if (lane == 63) p.nch[0] = nc;
}
__global__ void inv_perm_kernel(int B, Rows mod_sel, InvPlan p) {
  MPE_FOREGROUND();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  p.perm[p.cnt[mod_index(mod_sel, i)] + p.rank[i]] = i;
}

// ---------------------------------------------------------------------------------------------
// up-sweep / down-sweep on the Montgomery engine: one lane group per chunk
// ---------------------------------------------------------------------------------------------
// up: xm[pos] = Mont(x), pre[pos] = Mont(x_0 ... x_j) for the j-th item of the chunk; T[chunk] = the total (words)
template <class C>
__global__ void __launch_bounds__(64) inv_up_kernel(ModsetView ms, Rows A, InvPlan p, int maxch, uint32_t* __restrict__ xm,
                                                    uint32_t* __restrict__ pre, uint32_t* __restrict__ T) {
  MPE_FOREGROUND();
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int nch = p.nch[0];
  const int nslots = gridDim.x * C::GROUPS;
  const int trips = (maxch + nslots - 1) / nslots;
#pragma unroll 1
  for (int trip = 0; trip < trips; ++trip) {
    const int ch_raw = trip * nslots + blockIdx.x * C::GROUPS + ln.g;
    const bool active = ch_raw < nch;
    const int ch = active ? ch_raw : 0;
    const int mi = p.ch_mod[ch], start = p.ch_start[ch];
    // every group of the wave walks the longest chunk of the wave; shorter chunks repeat their last item
    int len = p.ch_len[ch], lmax = len;
    for (int o = C::TPI; o < 64; o <<= 1) { const int t = __shfl_xor(lmax, o); lmax = t > lmax ? t : lmax; }
    uint32_t n[C::L];
    load_owner<C>(n, ms.n_limbs + (size_t)mi * C::K, ln);
    const uint32_t n0inv = ms.n0inv[mi];
    uint32_t P[C::L], cur[C::L];
    load_owner<C>(P, ms.one_limbs + (size_t)mi * C::K, ln);
    const int nsteps = 2 * lmax + 1;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
      const int j = s >> 1, jc = j < len ? j : len - 1, pos = start + jc;
      const bool live = active && j < len;
      if (s == nsteps - 1) {                                   // leave the Montgomery domain: P * 1
#pragma unroll
        for (int i = 0; i < C::L; ++i) { cur[i] = P[i]; gl[ln.t * C::L + i] = (ln.t == 0 && i == 0) ? 1u : 0u; }
      } else if ((s & 1) == 0) {                               // Mont(x_j) = x_j * R^2 / R
        load_words_as_limbs<C>(cur, gl, row_of(A, p.perm[pos]), A.words, ln);
        copy_to_lds<C>(gl, ms.r2_limbs + (size_t)mi * C::K, ln);
      } else {                                                 // P <- P * Mont(x_j)   (multiplier already in LDS)
#pragma unroll
        for (int i = 0; i < C::L; ++i) cur[i] = P[i];
      }
      wave_lds_sync();
      uint32_t r[C::L];
      montmul<C>(r, cur, gl, n, n0inv, ln);
      wave_lds_sync();
      if (s == nsteps - 1) {
        reduce_once<C>(r, n, ln);
        store_limbs_as_words<C>(T + (size_t)ch * C::K32, gl, r, active, ln);
      } else if ((s & 1) == 0) {
        if (live) store_owner<C>(xm + (size_t)pos * C::K, r, ln);
        put_limbs<C>(gl, r, ln);                               // becomes the multiplier of the next step
      } else if (live) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) P[i] = r[i];
        store_owner<C>(pre + (size_t)pos * C::K, P, ln);
      }
    }
  }
}

// down: out[item] = x_j^-1 (words), ok[item]; chunks whose total was not invertible flag their items for the fallback
template <class C>
__global__ void __launch_bounds__(64) inv_down_kernel(ModsetView ms, InvPlan p, int maxch, const uint32_t* __restrict__ xm,
                                                      const uint32_t* __restrict__ pre, const uint32_t* __restrict__ Tinv,
                                                      const uint8_t* __restrict__ Tok, uint32_t* __restrict__ out,
                                                      uint8_t* __restrict__ ok, uint8_t* __restrict__ need_fallback) {
  MPE_FOREGROUND();
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int nch = p.nch[0];
  const int nslots = gridDim.x * C::GROUPS;
  const int trips = (maxch + nslots - 1) / nslots;
#pragma unroll 1
  for (int trip = 0; trip < trips; ++trip) {
    const int ch_raw = trip * nslots + blockIdx.x * C::GROUPS + ln.g;
    const bool active = ch_raw < nch;
    const int ch = active ? ch_raw : 0;
    const int mi = p.ch_mod[ch], start = p.ch_start[ch];
    int len = p.ch_len[ch], lmax = len;
    for (int o = C::TPI; o < 64; o <<= 1) { const int t = __shfl_xor(lmax, o); lmax = t > lmax ? t : lmax; }
    const bool tok = Tok[ch] != 0;
    if (active && !tok && ln.t0)
      for (int j = 0; j < len; ++j) need_fallback[p.perm[start + j]] = 1;
    uint32_t n[C::L];
    load_owner<C>(n, ms.n_limbs + (size_t)mi * C::K, ln);
    const uint32_t n0inv = ms.n0inv[mi];
    uint32_t inv[C::L], cur[C::L], y[C::L];
    // step 0: inv = Mont(T^-1);  then per item j = len-1 .. 0:  y = inv * P_{j-1};  out = y * 1;  inv = inv * Mont(x_j)
    const int nsteps = 1 + 3 * lmax;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
      const int jr = (s - 1) / 3, ph = (s - 1) % 3;            // jr-th item from the top
      const int j = len - 1 - jr, jc = j < 0 ? 0 : j, pos = start + jc;
      const bool live = active && tok && s > 0 && j >= 0;
      if (s == 0) {
        load_words_as_limbs<C>(cur, gl, Tinv + (size_t)ch * C::K32, 0, ln);
        copy_to_lds<C>(gl, ms.r2_limbs + (size_t)mi * C::K, ln);
      } else if (ph == 0) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) cur[i] = inv[i];
        copy_to_lds<C>(gl, jc > 0 ? pre + (size_t)(pos - 1) * C::K : ms.one_limbs + (size_t)mi * C::K, ln);
      } else if (ph == 1) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) { cur[i] = y[i]; gl[ln.t * C::L + i] = (ln.t == 0 && i == 0) ? 1u : 0u; }
      } else {
#pragma unroll
        for (int i = 0; i < C::L; ++i) cur[i] = inv[i];
        copy_to_lds<C>(gl, xm + (size_t)pos * C::K, ln);
      }
      wave_lds_sync();
      uint32_t r[C::L];
      montmul<C>(r, cur, gl, n, n0inv, ln);
      wave_lds_sync();
      if (s == 0) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) inv[i] = r[i];
      } else if (ph == 0) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) y[i] = r[i];
      } else if (ph == 1) {
        reduce_once<C>(r, n, ln);
        const int item = p.perm[pos];
        store_limbs_as_words<C>(out + (size_t)item * C::K32, gl, r, live, ln);
        if (live && ln.t0) ok[item] = 1;
      } else {
#pragma unroll
        for (int i = 0; i < C::L; ++i) inv[i] = r[i];
      }
    }
  }
}

static ModsetView inv_view_of(const mpe_modset* ms) {
  ModsetView v;
  v.n_limbs = ms->n_limbs; v.one_limbs = ms->one_limbs; v.r2_limbs = ms->r2_limbs; v.r2h_limbs = ms->r2h_limbs;
  v.n0inv = ms->n0inv; v.count = ms->count;
  return v;
}

static size_t modinv_ws_words(const mpe_modset* ms, int B) {
  const int K = ms->K, K32 = ms->bits / 32, maxch = B / 16 + (ms->count < B ? ms->count : B) + 2;     // chunks <= B / chunk + the moduli that have items
  return (size_t)B * (2 * K + 3) + (size_t)maxch * (2 * K32 + 8) + ms->count + 4096;
}

// out/ok: [B]; scratch comes from the context workspace (callers include modinv_ws_words in their reservation)
template <class C, typename LT>
static int launch_modinv_batched(mpe_ctx* ctx, const mpe_modset* ms, int B, Rows mod_sel, Rows a, uint32_t* out, uint8_t* ok,
                                 hipStream_t st) {
  // chunk = items inverted through ONE real inversion: 64 for throughput; a small batch is latency-bound, and the up / down
  // sweeps are `chunk` sequential multiplications, so it takes short chunks (more, but concurrent, wave gcds)
  const int CH = B <= ctx->par_items ? 16 : 64;
  const int nmod = ms->count, maxch = B / CH + (nmod < B ? nmod : B) + 2;
  InvPlan p;
  p.cnt = ws_array<int32_t>(ctx, nmod + 1);
  p.rank = ws_array<int32_t>(ctx, B);
  p.perm = ws_array<int32_t>(ctx, B);
  p.ch_start = ws_array<int32_t>(ctx, maxch);
  p.ch_len = ws_array<int32_t>(ctx, maxch);
  p.ch_mod = ws_array<int32_t>(ctx, maxch);
  p.nch = ws_array<int32_t>(ctx, 1);
  uint32_t* xm = ws_array<uint32_t>(ctx, (size_t)B * C::K);
  uint32_t* pre = ws_array<uint32_t>(ctx, (size_t)B * C::K);
  uint32_t* T = ws_array<uint32_t>(ctx, (size_t)maxch * C::K32);
  uint32_t* Tinv = ws_array<uint32_t>(ctx, (size_t)maxch * C::K32);
  uint8_t* Tok = ws_array<uint8_t>(ctx, maxch);
  uint8_t* need = ws_array<uint8_t>(ctx, B);
  if (!p.cnt || !p.rank || !p.perm || !p.ch_start || !p.ch_len || !p.ch_mod || !p.nch || !xm || !pre || !T || !Tinv || !Tok || !need) {
    mpe_set_error_msg("modinv: workspace under-reserved");
    return MPE_E_NOMEM;
  }
  (void)hipMemsetAsync(p.cnt, 0, (size_t)(nmod + 1) * 4, st);
  (void)hipMemsetAsync(need, 0, (size_t)B, st);
  (void)hipMemsetAsync(ok, 0, (size_t)B, st);
  (void)hipMemsetAsync(Tok, 0, (size_t)maxch, st);
  hipLaunchKernelGGL(inv_count_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, st, B, mod_sel, p);
  hipLaunchKernelGGL(inv_plan_kernel, dim3(1), dim3(64), 0, st, nmod, CH, p);
  hipLaunchKernelGGL(inv_perm_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, st, B, mod_sel, p);
  const int need_waves = (maxch + C::GROUPS - 1) / C::GROUPS;
  const int cap = ctx->cus * ctx->modexp_waves_per_cu;
  const int grid = need_waves < cap ? need_waves : cap;
  hipLaunchKernelGGL(inv_up_kernel<C>, dim3(grid), dim3(64), 0, st, inv_view_of(ms), a, p, maxch, xm, pre, T);
  hipLaunchKernelGGL(modinv_wave_kernel<LT>, dim3(maxch), dim3(64), 0, st, p.nch, ms->words, p.ch_mod, T, Tinv, Tok);
  hipLaunchKernelGGL(inv_down_kernel<C>, dim3(grid), dim3(64), 0, st, inv_view_of(ms), p, maxch, xm, pre, Tinv, Tok, out, ok, need);
  // exact per-item answers for chunks that contained a non-invertible element
  hipLaunchKernelGGL(modinv_lane_kernel<C::K32>, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, ms->words, mod_sel, a, need, out, ok);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("modinv (batched)", e); return MPE_E_HIP; }
  return MPE_OK;
}

static int launch_modinv(mpe_ctx* ctx, const mpe_modset* ms, int B, Rows mod_sel, Rows a, uint32_t* out, uint8_t* ok,
                         hipStream_t st) {
  if (B == 0) return MPE_OK;
  // A handful of items: the lane-serial kernel, one launch.  (Rounds 2-5 also sent every launch with "nearly one modulus per item" here —
  // nothing to batch — but the lane-serial binary gcd works in scratch memory and manages ~0.25 M inversions/s; chunks of ONE item through
  // the wave-cooperative gcd are ~7x that at scale: 16 384 sessions with a wallet each spent 74 % of their time in this kernel,
  // profiles/r06/kernel_stats_wallets_16384x16384.csv.)
  if (B <= 32 && ms->count > 1) {
    if (ms->bits == 4096)
      hipLaunchKernelGGL(modinv_lane_kernel<128>, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, ms->words, mod_sel, a,
                         (const uint8_t*)nullptr, out, ok);
    else
      hipLaunchKernelGGL(modinv_lane_kernel<64>, dim3(blocks_for(B, 64)), dim3(64), 0, st, B, ms->words, mod_sel, a,
                         (const uint8_t*)nullptr, out, ok);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mpe_set_error("modinv_lane_kernel", e); return MPE_E_HIP; }
    return MPE_OK;
  }
  if (ms->bits == 4096) return launch_modinv_batched<Cfg4096, uint64_t>(ctx, ms, B, mod_sel, a, out, ok, st);
  return launch_modinv_batched<Cfg2048, uint32_t>(ctx, ms, B, mod_sel, a, out, ok, st);
}

}  // namespace mpe
