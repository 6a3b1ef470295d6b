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
//
// `split` = S (1, 2, 4, 8, 16): the S lane groups of a wave that follow each other share ONE item.  A fixed-base power has no
// squarings — it is a product of nwin table rows — so the product can be cut into S runs of ~nwin / S rows, one per group, whose
// partial products (Montgomery forms, like the rows) are multiplied together in a tree through LDS: a serial chain of
// ceil(nwin / S) + log2(S) + 1 multiplications instead of nwin + 1 (217 windows at 13 bits: 218 -> 32 with S = 8).  A small launch
// lasts as long as ONE chain however few items it has (1 024 sessions: 2.2 - 5.7 ms per launch, a dozen launches per batch), so the
// host picks the largest S that still leaves one wave per SIMD (launch_fb_modexp); large launches keep S = 1, where every lane
// multiplies useful rows all the time.  Same residue: the factors are the same rows, multiplied in another order.
template <class C>
__global__ void __launch_bounds__(64) fb_modexp_kernel(int batch, ModsetView ms, Rows st_sel, int which,
                                                       const uint32_t* __restrict__ tab, int wb, Rows exps, int exp_words,
                                                       uint32_t* __restrict__ out, SchedArgs sched, int split) {
  const int FB_WB = wb, FB_TE = 1 << wb, FB_MAX_WINDOWS = fb_windows(wb);
  __shared__ __attribute__((aligned(16))) uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int S = split, per_wave = C::GROUPS / S;             // items per wave
  const int part = ln.g & (S - 1), sub = ln.g / S;
  const int nslots = gridDim.x * per_wave;
  const int nwin = (exp_words * 32 + FB_WB - 1) / FB_WB;
  const int cmax = (nwin + S - 1) / S;                       // rows of the longest run
  const int lo = part * nwin / S, cnt = (part + 1) * nwin / S - lo;
  WaveSched ws;                                          // mpe_sched.h: which unit this wave runs next
  ws.init(sched);
#pragma unroll 1
  for (;;) {
    int ubase;
    if (!ws.next(sched, batch, nslots, per_wave, ubase)) break;            // nothing left for this wave (wave-uniform)
    const int inst = ubase + sub;
    const bool active = inst < batch;
    const int idx = active ? inst : batch - 1;
    const int st = sel_index(st_sel, idx);
    const uint32_t* ex = row_of(exps, idx);
    const uint32_t* T = tab + (size_t)(2 * st + which) * FB_MAX_WINDOWS * FB_TE * C::K;
    const uint32_t* one_row = ms.one_limbs + (size_t)st * C::K;           // the form of 1: what a run multiplies by when it has no row left
    uint32_t n[C::L];
    load_owner<C>(n, ms.n_limbs + (size_t)st * C::K, ln);
    const uint32_t n0inv = ms.n0inv[st];
    auto row_ptr = [&](int k) -> const uint32_t* {           // k-th row of this group's run
      return k < cnt ? T + ((size_t)(lo + k) * FB_TE + fb_digit(ex, exp_words, lo + k, wb)) * C::K : one_row;
    };
    uint32_t cur[C::L];
    load_owner<C>(cur, row_ptr(0), ln);
    // steps 1..cmax-1: cur <- cur * row_k; then the tree over the S runs; last step: cur <- cur * 1.  Two table rows are in flight
    // while a third is multiplied (rows s+1 and s+2 travel during step s): one Montgomery product is ~1.4 us of VALU work, an HBM/L2
    // miss of a table row is about the same, so one row ahead left the tail of the fetch exposed (r04 A/B: -1.4 % on this kernel).
    // A row is K contiguous words (288 B at K = 72) and so is the multiplier in the group's LDS region: it moves as 16-byte loads
    // — the group's four lanes cover one 64-byte sector per instruction, each sector requested once (a word-by-word fetch asks
    // for every sector four times, 18 instructions per lane instead of 5: -3.4 %) — and 8-byte LDS stores (C::STRIDE is even).
    static_assert(C::K % 4 == 0 && C::STRIDE % 2 == 0, "wide row fetch needs 16-byte chunks and 8-byte aligned group regions");
    constexpr int CHUNKS = C::K / 4, PER = (CHUNKS + C::TPI - 1) / C::TPI;
    uint4 odd[PER], even[PER];                       // the rows of the next odd and the next even step
    auto fetch = [&](uint4 (&dst)[PER], int k) {
      const uint4* src = reinterpret_cast<const uint4*>(row_ptr(k));
#pragma unroll
      for (int c = 0; c < PER; ++c) {
        const int ch = ln.t + C::TPI * c;
        if ((c + 1) * C::TPI <= CHUNKS || ch < CHUNKS) dst[c] = src[ch];
      }
    };
    auto put = [&](const uint4 (&row)[PER]) {
      uint2* l2 = reinterpret_cast<uint2*>(gl);
#pragma unroll
      for (int c = 0; c < PER; ++c) {
        const int ch = ln.t + C::TPI * c;
        if ((c + 1) * C::TPI <= CHUNKS || ch < CHUNKS) {
          l2[2 * ch] = make_uint2(row[c].x, row[c].y);
          l2[2 * ch + 1] = make_uint2(row[c].z, row[c].w);
        }
      }
    };
    auto mul_by = [&](const uint32_t* region) {      // cur <- cur * (the K limbs at `region` of the wave's LDS)
      wave_lds_sync();
      uint32_t r[C::L];
      montmul<C>(r, cur, region, n, n0inv, ln);
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < C::L; ++k) cur[k] = r[k];
    };
    // ONE chain of cmax - 1 + log2(S) + 1 multiplications, two call sites of the multiplication (a third and fourth cost the
    // kernel its second wave per SIMD: 269 registers): step s multiplies by a table row (s < cmax), then — the tree — by the partial
    // product of the run h = 1, 2, 4 ... groups further on (every group publishes its own and multiplies: one instruction stream
    // per wave; what the groups that are not a multiple of 2h compute is never read again), and last by 1 (out of the Montgomery form)
    int lg = 0;
    while ((1 << lg) < S) ++lg;
    const int total = cmax - 1 + lg + 1;
    auto step = [&](int sidx, uint4 (&buf)[PER]) {
      const uint32_t* region = gl;
      if (sidx < cmax) {
        put(buf);
        if (sidx + 2 < cmax) fetch(buf, sidx + 2);
      } else if (sidx < cmax + lg) {
        const int h = 1 << (sidx - cmax);
#pragma unroll
        for (int k = 0; k < C::L; ++k) gl[ln.t * C::L + k] = cur[k];
        region = lds + ((ln.g + h < C::GROUPS) ? ln.g + h : ln.g) * C::STRIDE;
      } else {
#pragma unroll
        for (int k = 0; k < C::L; ++k) gl[ln.t * C::L + k] = (ln.t == 0 && k == 0) ? 1u : 0u;
      }
      mul_by(region);
    };
    if (cmax > 1) fetch(odd, 1);
    if (cmax > 2) fetch(even, 2);
#pragma unroll 1
    for (int sidx = 1; sidx <= total; sidx += 2) {
      step(sidx, odd);
      if (sidx + 1 <= total) step(sidx + 1, even);
    }
    reduce_once<C>(cur, n, ln);
    store_limbs_as_words<C>(out + (size_t)idx * C::K32, gl, cur, active && part == 0, ln);
  }
}

}  // namespace mpe
