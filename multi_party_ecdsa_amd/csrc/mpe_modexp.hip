// Batched modular exponentiation / multiplication kernels and their C-ABI (include/mpecdsa_hip.h).
// gfx950 only.  See mpe_bigint.h for the arithmetic design.
#include "mpe_internal.h"

namespace mpe {

// ---------------------------------------------------------------------------------------------
// modulus set-up kernel: one group per modulus
//   n limbs, n0inv = -n^-1 mod 2^W, one = R mod n, r2 = R^2 mod n     (R = 2^(W*K))
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) modset_setup_kernel(int count, const uint32_t* __restrict__ moduli,
                                                          uint32_t* __restrict__ n_limbs,
                                                          uint32_t* __restrict__ one_limbs,
                                                          uint32_t* __restrict__ r2_limbs,
                                                          uint32_t* __restrict__ n0inv_out) {
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const bool active = slot < count;
  const int idx = active ? slot : count - 1;

  stage_words<C>(gl, moduli + (size_t)idx * C::K32, ln);
  wave_lds_sync();
  uint32_t n[C::L];
  limbs_from_words<C>(n, gl, ln);
  // -n^-1 mod 2^32 by Newton iteration on the low word, then truncated to W bits
  const uint32_t n0 = gl[0];
  uint32_t inv = n0;                       // correct to 3 bits for odd n0
#pragma unroll
  for (int i = 0; i < 5; ++i) inv *= 2u - n0 * inv;
  const uint32_t n0inv = (0u - inv) & C::MASK;
  // bit length of n (lane 0 of the group scans the staged words)
  int bl = 0;
  if (ln.t0) {
    for (int q = C::K32 - 1; q >= 0; --q) {
      const uint32_t w = gl[q];
      if (w != 0) { bl = q * 32 + (32 - __builtin_clz(w)); break; }
    }
  }
  bl = (int)bcast0<C::TPI>((uint32_t)bl);
  wave_lds_sync();

  // x = 2^(bl-1) < n, then double (mod n) up to 2^(W*K) mod n
  int64_t x[C::L];
#pragma unroll
  for (int i = 0; i < C::L; ++i) {
    const int p = ln.t * C::L + i;
    x[i] = (p == (bl - 1) / C::W) ? ((int64_t)1 << ((bl - 1) % C::W)) : 0;
  }
  const int doublings = C::W * C::K - (bl - 1);
#pragma unroll 1
  for (int d = 0; d < doublings; ++d) {
#pragma unroll
    for (int i = 0; i < C::L; ++i) x[i] *= 2;
    full_normalize<C>(x, ln);
    if (cmp_ge<C>(x, n, ln)) {
#pragma unroll
      for (int i = 0; i < C::L; ++i) x[i] -= (int64_t)n[i];
      full_normalize<C>(x, ln);
    }
  }
  uint32_t one[C::L];
#pragma unroll
  for (int i = 0; i < C::L; ++i) one[i] = (uint32_t)x[i];

  // r2 = Mont(2^(W*K)): square-and-double in the Montgomery domain starting from Mont(2)
  uint32_t y[C::L];
  auto dbl = [&](uint32_t (&v)[C::L]) {
    int64_t z[C::L];
#pragma unroll
    for (int i = 0; i < C::L; ++i) z[i] = 2 * (int64_t)v[i];
    full_normalize<C>(z, ln);
    if (cmp_ge<C>(z, n, ln)) {
#pragma unroll
      for (int i = 0; i < C::L; ++i) z[i] -= (int64_t)n[i];
      full_normalize<C>(z, ln);
    }
#pragma unroll
    for (int i = 0; i < C::L; ++i) v[i] = (uint32_t)z[i];
  };
#pragma unroll
  for (int i = 0; i < C::L; ++i) y[i] = one[i];
  dbl(y);                                         // Mont(2^1)
  constexpr int E = C::W * C::K;
  constexpr int TOP = 31 - __builtin_clz((unsigned)E);
#pragma unroll 1
  for (int b = TOP - 1; b >= 0; --b) {
    put_limbs<C>(gl, y, ln);
    wave_lds_sync();
    uint32_t r[C::L];
    montmul<C>(r, y, gl, n, n0inv, ln);
    wave_lds_sync();
    reduce_once<C>(r, n, ln);
#pragma unroll
    for (int i = 0; i < C::L; ++i) y[i] = r[i];
    if ((E >> b) & 1) dbl(y);
  }
  if (active) {
#pragma unroll
    for (int i = 0; i < C::L; ++i) {
      const size_t o = (size_t)idx * C::K + ln.t * C::L + i;
      n_limbs[o] = n[i];
      one_limbs[o] = one[i];
      r2_limbs[o] = y[i];
    }
    if (ln.t0) n0inv_out[idx] = n0inv;
  }
}

// ---------------------------------------------------------------------------------------------
// helpers shared by the modexp / modmul kernels
// ---------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void load_owner(uint32_t (&v)[C::L], const uint32_t* __restrict__ src, const Lane& ln) {
#pragma unroll
  for (int i = 0; i < C::L; ++i) v[i] = src[ln.t * C::L + i];
}
template <class C>
__device__ __forceinline__ void store_owner(uint32_t* __restrict__ dst, const uint32_t (&v)[C::L], const Lane& ln) {
#pragma unroll
  for (int i = 0; i < C::L; ++i) dst[ln.t * C::L + i] = v[i];
}
// global limb array (K words) -> the group's LDS "b" region, coalesced within the group
template <class C>
__device__ __forceinline__ void copy_to_lds(uint32_t* gl, const uint32_t* __restrict__ src, const Lane& ln) {
#pragma unroll
  for (int i = 0; i < C::L; ++i) gl[ln.t + C::TPI * i] = src[ln.t + C::TPI * i];
}
struct ModsetView {
  const uint32_t* n_limbs;
  const uint32_t* one_limbs;
  const uint32_t* r2_limbs;
  const uint32_t* n0inv;
  int count;
};

// ---------------------------------------------------------------------------------------------
// modexp kernel: persistent waves, each group walks the batch with a grid stride
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) modexp_kernel(int batch, ModsetView ms, const int32_t* __restrict__ mod_idx,
                                                    const uint32_t* __restrict__ base,
                                                    const uint32_t* __restrict__ exps, int exp_words,
                                                    uint32_t* __restrict__ out, uint32_t* __restrict__ tables) {
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const int nslots = gridDim.x * C::GROUPS;
  uint32_t* tab = tables + (size_t)slot * 16 * C::K;     // this group's 16-entry window table
  const int trips = (batch + nslots - 1) / nslots;
  const int nwin = exp_words * 8;
  // One Montgomery multiplication per step; the step index alone (wave-uniform) decides where the
  // multiplier comes from and where the product goes, so montmul is instantiated exactly once:
  //   step 0            : cur = base * R^2            -> Mont(base) = tab[1]
  //   step 1..14        : cur = cur * Mont(base)      -> tab[2..15]
  //   then per window   : 4 squarings, 1 multiplication by tab[window]
  //   last step         : cur = cur * 1               -> leaves the Montgomery domain
  const int nsteps = 15 + 5 * (nwin - 1) + 1;

#pragma unroll 1
  for (int trip = 0; trip < trips; ++trip) {
    const int inst = trip * nslots + slot;
    const bool active = inst < batch;
    const int idx = active ? inst : batch - 1;
    const int mi = mod_idx ? mod_idx[idx] : (ms.count == 1 ? 0 : idx);
    const uint32_t* ex = exps + (size_t)idx * exp_words;

    uint32_t n[C::L];
    load_owner<C>(n, ms.n_limbs + (size_t)mi * C::K, ln);
    const uint32_t n0inv = ms.n0inv[mi];

    stage_words<C>(gl, base + (size_t)idx * C::K32, ln);
    wave_lds_sync();
    uint32_t cur[C::L];
    limbs_from_words<C>(cur, gl, ln);
    wave_lds_sync();
    {
      uint32_t one[C::L];
      load_owner<C>(one, ms.one_limbs + (size_t)mi * C::K, ln);
      store_owner<C>(tab, one, ln);                         // tab[0] = Mont(1)
    }

#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
      // ---- multiplier -> LDS ----
      if (step == 0) {
        copy_to_lds<C>(gl, ms.r2_limbs + (size_t)mi * C::K, ln);
      } else if (step == 1) {
        put_limbs<C>(gl, cur, ln);                          // Mont(base) stays in LDS for steps 1..14
      } else if (step >= 15 && step < nsteps - 1) {
        const int k = step - 15;
        const int wi = nwin - 2 - k / 5;
        if (k % 5 == 4) {
          const uint32_t w = (ex[wi >> 3] >> ((wi & 7) * 4)) & 15u;
          copy_to_lds<C>(gl, tab + (size_t)w * C::K, ln);
        } else {
          put_limbs<C>(gl, cur, ln);                        // squaring
        }
      } else if (step == nsteps - 1) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) gl[ln.t * C::L + i] = (ln.t == 0 && i == 0) ? 1u : 0u;
      }
      wave_lds_sync();
      uint32_t r[C::L];
      montmul<C>(r, cur, gl, n, n0inv, ln);
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < C::L; ++i) cur[i] = r[i];
      // ---- product -> window table ----
      if (step < 15) {
        store_owner<C>(tab + (size_t)(step + 1) * C::K, cur, ln);
        if (step == 14) {
          // table complete: start the ladder from the top window
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          const uint32_t w = (ex[(nwin - 1) >> 3] >> (((nwin - 1) & 7) * 4)) & 15u;
          load_owner<C>(cur, tab + (size_t)w * C::K, ln);
        }
      }
    }
    // canonical residue -> interface words
    reduce_once<C>(cur, n, ln);
    put_limbs<C>(gl, cur, ln);
    if (ln.t0) { gl[C::K] = 0; gl[C::K + 1] = 0; }
    wave_lds_sync();
    if (active)
      for (int q = ln.t; q < C::K32; q += C::TPI) out[(size_t)idx * C::K32 + q] = word_from_limbs<C>(gl, q);
    wave_lds_sync();
  }
}

// ---------------------------------------------------------------------------------------------
// modmul kernel: out = a*b mod n   (two Montgomery multiplications: by b, then by R^2)
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) modmul_kernel(int batch, ModsetView ms, const int32_t* __restrict__ mod_idx,
                                                    const uint32_t* __restrict__ A, const uint32_t* __restrict__ B,
                                                    uint32_t* __restrict__ out) {
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const int nslots = gridDim.x * C::GROUPS;
  const int trips = (batch + nslots - 1) / nslots;
#pragma unroll 1
  for (int trip = 0; trip < trips; ++trip) {
    const int inst = trip * nslots + slot;
    const bool active = inst < batch;
    const int idx = active ? inst : batch - 1;
    const int mi = mod_idx ? mod_idx[idx] : (ms.count == 1 ? 0 : idx);
    uint32_t n[C::L];
    load_owner<C>(n, ms.n_limbs + (size_t)mi * C::K, ln);
    const uint32_t n0inv = ms.n0inv[mi];

    stage_words<C>(gl, A + (size_t)idx * C::K32, ln);
    wave_lds_sync();
    uint32_t a[C::L];
    limbs_from_words<C>(a, gl, ln);
    wave_lds_sync();
    // b (interface words) -> limbs -> LDS limb order
    stage_words<C>(gl, B + (size_t)idx * C::K32, ln);
    wave_lds_sync();
    uint32_t b[C::L];
    limbs_from_words<C>(b, gl, ln);
    wave_lds_sync();
    put_limbs<C>(gl, b, ln);
    wave_lds_sync();
    uint32_t t[C::L];
    montmul<C>(t, a, gl, n, n0inv, ln);              // a*b/R
    wave_lds_sync();
    copy_to_lds<C>(gl, ms.r2_limbs + (size_t)mi * C::K, ln);
    wave_lds_sync();
    uint32_t u[C::L];
    montmul<C>(u, t, gl, n, n0inv, ln);              // a*b mod n (lazy, < 2n)
    wave_lds_sync();
    reduce_once<C>(u, n, ln);
    put_limbs<C>(gl, u, ln);
    if (ln.t0) { gl[C::K] = 0; gl[C::K + 1] = 0; }
    wave_lds_sync();
    if (active)
      for (int q = ln.t; q < C::K32; q += C::TPI) out[(size_t)idx * C::K32 + q] = word_from_limbs<C>(gl, q);
    wave_lds_sync();
  }
}

}  // namespace mpe

// =============================================================================================
// C-ABI
// =============================================================================================
using namespace mpe;

thread_local std::string g_last_error;
void mpe_set_error(const char* what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}

extern "C" int mpe_modset_destroy(mpe_modset* ms);

template <class C>
static int modset_create_impl(mpe_ctx* ctx, int count, const uint32_t* d_moduli, mpe_modset** out, hipStream_t st) {
  mpe_modset* ms = new (std::nothrow) mpe_modset();
  if (!ms) return MPE_E_NOMEM;
  ms->bits = C::BITS;
  ms->count = count;
  ms->K = C::K;
  const size_t words = (size_t)count * C::K;
  const size_t total = (3 * words + (size_t)count) * sizeof(uint32_t);
  hipError_t e = hipMalloc(&ms->blob, total);
  if (e != hipSuccess) { delete ms; mpe_set_error("hipMalloc(modset)", e); return MPE_E_NOMEM; }
  uint32_t* p = (uint32_t*)ms->blob;
  ms->n_limbs = p;
  ms->one_limbs = p + words;
  ms->r2_limbs = p + 2 * words;
  ms->n0inv = p + 3 * words;
  const int blocks = (count + C::GROUPS - 1) / C::GROUPS;
  hipLaunchKernelGGL(modset_setup_kernel<C>, dim3(blocks), dim3(64), 0, st, count, d_moduli, ms->n_limbs,
                     ms->one_limbs, ms->r2_limbs, ms->n0inv);
  e = hipGetLastError();
  if (e != hipSuccess) { mpe_modset_destroy(ms); mpe_set_error("modset_setup_kernel", e); return MPE_E_HIP; }
  (void)ctx;
  *out = ms;
  return MPE_OK;
}

static ModsetView view_of(const mpe_modset* ms) {
  ModsetView v;
  v.n_limbs = ms->n_limbs; v.one_limbs = ms->one_limbs; v.r2_limbs = ms->r2_limbs; v.n0inv = ms->n0inv;
  v.count = ms->count;
  return v;
}

// persistent grid: at most the resident-wave capacity; when the batch needs several trips the
// grid shrinks to the smallest one that still finishes in that many trips (no half-empty tail)
template <class C>
static int grid_for(const mpe_ctx* ctx, int batch, int waves_per_cu) {
  const int need = (batch + C::GROUPS - 1) / C::GROUPS;
  const int cap = ctx->cus * waves_per_cu;
  if (need <= cap) return need;
  const int trips = (need + cap - 1) / cap;
  return (need + trips - 1) / trips;
}

template <class C>
static int modexp_impl(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx, const uint32_t* d_base,
                       const uint32_t* d_exp, int exp_words, uint32_t* d_out, hipStream_t st) {
  const int grid = grid_for<C>(ctx, batch, ctx->modexp_waves_per_cu);
  const size_t need = (size_t)grid * C::GROUPS * 16 * C::K * sizeof(uint32_t);
  if (need > ctx->tables_bytes) {
    if (ctx->tables) { (void)hipStreamSynchronize(st); (void)hipFree(ctx->tables); ctx->tables = nullptr; ctx->tables_bytes = 0; }
    hipError_t e = hipMalloc(&ctx->tables, need);
    if (e != hipSuccess) { mpe_set_error("hipMalloc(window tables)", e); return MPE_E_NOMEM; }
    ctx->tables_bytes = need;
  }
  hipLaunchKernelGGL(modexp_kernel<C>, dim3(grid), dim3(64), 0, st, batch, view_of(ms), d_mod_idx, d_base, d_exp,
                     exp_words, d_out, (uint32_t*)ctx->tables);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("modexp_kernel", e); return MPE_E_HIP; }
  ctx->last.waves = grid;
  ctx->last.ints_per_wave = C::GROUPS;
  ctx->last.limbs = C::K;
  ctx->last.limb_bits = C::W;
  ctx->last.lds_bytes_per_wave = C::LDS_WORDS * 4;
  ctx->last.table_scratch_bytes = need;
  return MPE_OK;
}

template <class C>
static int modmul_impl(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx, const uint32_t* d_a,
                       const uint32_t* d_b, uint32_t* d_out, hipStream_t st) {
  const int grid = grid_for<C>(ctx, batch, 32);
  hipLaunchKernelGGL(modmul_kernel<C>, dim3(grid), dim3(64), 0, st, batch, view_of(ms), d_mod_idx, d_a, d_b, d_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("modmul_kernel", e); return MPE_E_HIP; }
  return MPE_OK;
}

extern "C" {

const char* mpe_version(void) { return "mpecdsa-hip 0.1.0 (gfx950)"; }
const char* mpe_last_error(void) { return g_last_error.c_str(); }

int mpe_ctx_create(mpe_ctx** out, int device) {
  if (!out) return MPE_E_ARG;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { mpe_set_error("hipSetDevice", e); return MPE_E_HIP; }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { mpe_set_error("hipGetDeviceProperties", e); return MPE_E_HIP; }
  mpe_ctx* c = new (std::nothrow) mpe_ctx();
  if (!c) return MPE_E_NOMEM;
  c->device = device;
  c->cus = prop.multiProcessorCount;
  *out = c;
  return MPE_OK;
}

int mpe_ctx_destroy(mpe_ctx* ctx) {
  if (!ctx) return MPE_E_ARG;
  if (ctx->tables) (void)hipFree(ctx->tables);
  delete ctx;
  return MPE_OK;
}

int mpe_sync(mpe_ctx* ctx, void* stream) {
  (void)ctx;
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) { mpe_set_error("hipStreamSynchronize", e); return MPE_E_HIP; }
  return MPE_OK;
}

int mpe_last_launch_info(const mpe_ctx* ctx, mpe_launch_info* out) {
  if (!ctx || !out) return MPE_E_ARG;
  *out = ctx->last;
  return MPE_OK;
}

int mpe_modset_count(const mpe_modset* ms) { return ms ? ms->count : MPE_E_ARG; }
int mpe_modset_bits(const mpe_modset* ms) { return ms ? ms->bits : MPE_E_ARG; }

int mpe_modset_destroy(mpe_modset* ms) {
  if (!ms) return MPE_E_ARG;
  if (ms->blob) (void)hipFree(ms->blob);
  delete ms;
  return MPE_OK;
}

int mpe_modset_create(mpe_ctx* ctx, int bits, int count, const uint32_t* d_moduli, mpe_modset** out, void* stream) {
  if (!ctx || !out || !d_moduli || count <= 0) return MPE_E_ARG;
  if (bits == 4096) return modset_create_impl<Cfg4096>(ctx, count, d_moduli, out, (hipStream_t)stream);
  if (bits == 2048) return modset_create_impl<Cfg2048>(ctx, count, d_moduli, out, (hipStream_t)stream);
  return MPE_E_ARG;
}

int mpe_modexp(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx, const uint32_t* d_base,
               const uint32_t* d_exp, int exp_words, uint32_t* d_out, void* stream) {
  if (!ctx || !ms || !d_base || !d_exp || !d_out || batch < 0 || exp_words <= 0) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  if (!d_mod_idx && ms->count != 1 && ms->count < batch) return MPE_E_ARG;
  if (ms->bits == 4096) return modexp_impl<Cfg4096>(ctx, ms, batch, d_mod_idx, d_base, d_exp, exp_words, d_out, (hipStream_t)stream);
  if (ms->bits == 2048) return modexp_impl<Cfg2048>(ctx, ms, batch, d_mod_idx, d_base, d_exp, exp_words, d_out, (hipStream_t)stream);
  return MPE_E_ARG;
}

int mpe_modmul(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx, const uint32_t* d_a,
               const uint32_t* d_b, uint32_t* d_out, void* stream) {
  if (!ctx || !ms || !d_a || !d_b || !d_out || batch < 0) return MPE_E_ARG;
  if (batch == 0) return MPE_OK;
  if (!d_mod_idx && ms->count != 1 && ms->count < batch) return MPE_E_ARG;
  if (ms->bits == 4096) return modmul_impl<Cfg4096>(ctx, ms, batch, d_mod_idx, d_a, d_b, d_out, (hipStream_t)stream);
  if (ms->bits == 2048) return modmul_impl<Cfg2048>(ctx, ms, batch, d_mod_idx, d_a, d_b, d_out, (hipStream_t)stream);
  return MPE_E_ARG;
}

}  // extern "C"
