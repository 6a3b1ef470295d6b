// Fixed-base exponentiation for the per-key bases h1, h2 of the DLogStatement tables.
//
// Almost every mod-N~ exponentiation on the GG20 path has base h1 or h2 of a statement
// (src/utilities/mta/range_proofs.rs:52,56-57,129-131; src/utilities/zk_pdl_with_slack/mod.rs:79-100,159-165),
// and those bases are fixed per key.  With a table T[i][d] = h^(d * 2^(wb i)) in HBM an exponentiation is just
// one Montgomery multiplication per wb-bit window: E/wb multiplications and no squarings, instead of E
// squarings + E/4 multiplications.  wb = 13 (the default): 220 windows x 8192 entries x 288 B = 0.5 GB per base —
// nothing next to 288 GB; each multiplication reads one 288-byte row (prefetched one step ahead).  Measured on
// one box, signatures/s at wb = 8 / 10 / 12 / 13 / 14: 16 151 / 16 408 / 16 555 / 16 590 / 16 669; wb = 16 (3.3 GB
// per base, 20 GB of randomly read rows) is slower than 8: the TLBs thrash.
// The value is the same residue mpz_powm returns.  The operation sequence depends only on exp_words.
#pragma once
#include "mpe_internal.h"

namespace mpe {

// window width wb: a property of a statement set (chosen when its tables are built; windows may straddle words)
constexpr int FB_EXP_BITS = 89 * 32;                            // exponents up to 89 words (s2, s3 < 2^2817)
__host__ __device__ inline int fb_windows(int wb) { return (FB_EXP_BITS + wb - 1) / wb; }

__device__ __forceinline__ uint32_t fb_digit(const uint32_t* __restrict__ ex, int exp_words, int i, int wb) {
  const int bitpos = i * wb, word = bitpos >> 5, sh = bitpos & 31;
  uint32_t v = ex[word] >> sh;
  if (sh + wb > 32 && word + 1 < exp_words) v |= ex[word + 1] << (32 - sh);
  return v & ((1u << wb) - 1u);
}

// Stage 1, one lane group per (statement, base): the window bases.
//   tab[pair][i][0] = Mont(1), tab[pair][i][1] = Mont(h^(2^(wb i)))       (canonical residues)
template <class C>
__global__ void __launch_bounds__(64) fb_bases_kernel(int npairs, ModsetView ms, const uint32_t* __restrict__ h1,
                                                      const uint32_t* __restrict__ h2, int wb, uint32_t* __restrict__ tab) {
  const int FB_WB = wb, FB_TE = 1 << wb, FB_MAX_WINDOWS = fb_windows(wb);
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const bool active = slot < npairs;
  const int pair = active ? slot : npairs - 1;
  const int st = pair >> 1;
  const uint32_t* hw = ((pair & 1) ? h2 : h1) + (size_t)st * C::K32;
  uint32_t n[C::L];
  load_owner<C>(n, ms.n_limbs + (size_t)st * C::K, ln);
  const uint32_t n0inv = ms.n0inv[st];
  uint32_t one[C::L], cur[C::L];
  load_owner<C>(one, ms.one_limbs + (size_t)st * C::K, ln);
  load_words_as_limbs<C>(cur, gl, hw, 0, ln);
  // step -1: cur = Mont(h); then FB_WB squarings per window
  const int nsteps = (FB_MAX_WINDOWS - 1) * FB_WB;
#pragma unroll 1
  for (int s = -1; s < nsteps; ++s) {
    if (s >= 0 && s % FB_WB == 0 && active) {
      uint32_t* row = tab + ((size_t)pair * FB_MAX_WINDOWS + s / FB_WB) * FB_TE * C::K;
      store_owner<C>(row, one, ln);
      store_owner<C>(row + C::K, cur, ln);
    }
    if (s < 0) copy_to_lds<C>(gl, ms.r2_limbs + (size_t)st * C::K, ln);
    else put_limbs<C>(gl, cur, ln);
    wave_lds_sync();
    uint32_t r[C::L];
    montmul<C>(r, cur, gl, n, n0inv, ln);
    wave_lds_sync();
    reduce_once<C>(r, n, ln);                        // keep table entries canonical
#pragma unroll
    for (int k = 0; k < C::L; ++k) cur[k] = r[k];
  }
  if (active) {
    uint32_t* row = tab + ((size_t)pair * FB_MAX_WINDOWS + (FB_MAX_WINDOWS - 1)) * FB_TE * C::K;
    store_owner<C>(row, one, ln);
    store_owner<C>(row + C::K, cur, ln);
  }
}

// Stage 2, one lane group per (statement, base, window): tab[..][d] = tab[..][d-1] * tab[..][1], d = 2..TE-1
template <class C>
__global__ void __launch_bounds__(64) fb_fill_kernel(int nrows, ModsetView ms, int wb, uint32_t* __restrict__ tab) {
  const int FB_TE = 1 << wb, FB_MAX_WINDOWS = fb_windows(wb);
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const bool active = slot < nrows;
  const int rowi = active ? slot : nrows - 1;                 // = pair * FB_MAX_WINDOWS + window
  const int st = (rowi / FB_MAX_WINDOWS) >> 1;
  uint32_t* row = tab + (size_t)rowi * FB_TE * C::K;
  uint32_t n[C::L];
  load_owner<C>(n, ms.n_limbs + (size_t)st * C::K, ln);
  const uint32_t n0inv = ms.n0inv[st];
  uint32_t cur[C::L];
  load_owner<C>(cur, row + C::K, ln);
  put_limbs<C>(gl, cur, ln);                                   // the window base stays in LDS
  wave_lds_sync();
#pragma unroll 1
  for (int d = 2; d < FB_TE; ++d) {
    uint32_t r[C::L];
    montmul<C>(r, cur, gl, n, n0inv, ln);
    reduce_once<C>(r, n, ln);
#pragma unroll
    for (int k = 0; k < C::L; ++k) cur[k] = r[k];
    if (active) store_owner<C>(row + (size_t)d * C::K, cur, ln);
  }
}

// out[i] = h^exp[i] mod N~ for h = h1 (which = 0) or h2 (which = 1) of statement st(i)
template <class C>
__global__ void __launch_bounds__(64) fb_modexp_kernel(int batch, ModsetView ms, Rows st_sel, int which,
                                                       const uint32_t* __restrict__ tab, int wb, Rows exps, int exp_words,
                                                       uint32_t* __restrict__ out, SchedArgs sched) {
  const int FB_WB = wb, FB_TE = 1 << wb, FB_MAX_WINDOWS = fb_windows(wb);
  __shared__ __attribute__((aligned(16))) uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const int nslots = gridDim.x * C::GROUPS;
  const int nwin = (exp_words * 32 + FB_WB - 1) / FB_WB;
  WaveSched ws;                                          // mpe_sched.h: which unit this wave runs next
  ws.init(sched);
#pragma unroll 1
  for (;;) {
    int ubase;
    if (!ws.next(sched, batch, nslots, C::GROUPS, ubase)) break;           // nothing left for this wave (wave-uniform)
    const int inst = ubase + ln.g;
    const bool active = inst < batch;
    const int idx = active ? inst : batch - 1;
    const int st = sel_index(st_sel, idx);
    const uint32_t* ex = row_of(exps, idx);
    const uint32_t* T = tab + (size_t)(2 * st + which) * FB_MAX_WINDOWS * FB_TE * C::K;
    uint32_t n[C::L];
    load_owner<C>(n, ms.n_limbs + (size_t)st * C::K, ln);
    const uint32_t n0inv = ms.n0inv[st];
    uint32_t cur[C::L];
    load_owner<C>(cur, T + (size_t)fb_digit(ex, exp_words, 0, wb) * C::K, ln);
    // steps 1..nwin-1: cur <- cur * T[i][digit_i]; step nwin: cur <- cur * 1.  Two table rows are in flight while a third is
    // multiplied (rows s+1 and s+2 travel during step s): one Montgomery product is ~1.4 us of VALU work, an HBM/L2 miss of a
    // table row is about the same, so one row ahead left the tail of the fetch exposed (r04 A/B: -1.4 % on this kernel).
    // A row is K contiguous words (288 B at K = 72) and so is the multiplier in the group's LDS region: it moves as 16-byte loads
    // — the group's four lanes cover one 64-byte sector per instruction, each sector requested once (a word-by-word fetch asks
    // for every sector four times, 18 instructions per lane instead of 5: -3.4 %) — and 8-byte LDS stores (C::STRIDE is even).
    static_assert(C::K % 4 == 0 && C::STRIDE % 2 == 0, "wide row fetch needs 16-byte chunks and 8-byte aligned group regions");
    constexpr int CHUNKS = C::K / 4, PER = (CHUNKS + C::TPI - 1) / C::TPI;
    uint4 odd[PER], even[PER];                       // the rows of the next odd and the next even step
    auto fetch = [&](uint4 (&dst)[PER], int i) {
      const uint4* src = reinterpret_cast<const uint4*>(T + ((size_t)i * FB_TE + fb_digit(ex, exp_words, i, wb)) * C::K);
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int ch = ln.t + C::TPI * k;
        if ((k + 1) * C::TPI <= CHUNKS || ch < CHUNKS) dst[k] = src[ch];
      }
    };
    auto put = [&](const uint4 (&row)[PER]) {
      uint2* l2 = reinterpret_cast<uint2*>(gl);
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int ch = ln.t + C::TPI * k;
        if ((k + 1) * C::TPI <= CHUNKS || ch < CHUNKS) {
          l2[2 * ch] = make_uint2(row[k].x, row[k].y);
          l2[2 * ch + 1] = make_uint2(row[k].z, row[k].w);
        }
      }
    };
    auto put_one = [&]() {
#pragma unroll
      for (int k = 0; k < C::L; ++k) gl[ln.t * C::L + k] = (ln.t == 0 && k == 0) ? 1u : 0u;
    };
    auto mul_by_lds = [&]() {
      wave_lds_sync();
      uint32_t r[C::L];
      montmul<C>(r, cur, gl, n, n0inv, ln);
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < C::L; ++k) cur[k] = r[k];
    };
    if (nwin > 1) fetch(odd, 1);
    if (nwin > 2) fetch(even, 2);
#pragma unroll 1
    for (int s = 1; s <= nwin; s += 2) {
      if (s < nwin) { put(odd); if (s + 2 < nwin) fetch(odd, s + 2); } else put_one();
      mul_by_lds();
      if (s + 1 <= nwin) {
        if (s + 1 < nwin) { put(even); if (s + 3 < nwin) fetch(even, s + 3); } else put_one();
        mul_by_lds();
      }
    }
    reduce_once<C>(cur, n, ln);
    store_limbs_as_words<C>(out + (size_t)idx * C::K32, gl, cur, active, ln);
  }
}

}  // namespace mpe
