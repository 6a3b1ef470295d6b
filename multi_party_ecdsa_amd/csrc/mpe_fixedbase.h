// Fixed-base exponentiation for the per-key bases h1, h2 of the DLogStatement tables.
//
// Almost every mod-N~ exponentiation on the GG20 path has base h1 or h2 of a statement
// (src/utilities/mta/range_proofs.rs:52,56-57,129-131; src/utilities/zk_pdl_with_slack/mod.rs:79-100,159-165),
// and those bases are fixed per key.  With a table T[i][d] = h^(d * 16^i) in HBM (712 windows x 16 entries x
// 288 B = 3.3 MB per base — nothing next to 288 GB) an exponentiation is just one Montgomery multiplication
// per 4-bit window: E/4 multiplications and no squarings, instead of E squarings + E/4 multiplications.
// The value is the same residue mpz_powm returns.  The operation sequence depends only on exp_words.
#pragma once
#include "mpe_internal.h"

namespace mpe {

constexpr int FB_MAX_WINDOWS = 89 * 8;     // exponents up to 89 words (s2, s3 < 2^2817)

// one lane group per (statement, base): tab[(pair * FB_MAX_WINDOWS + i) * 16 + d][K] = Mont(h^(d 16^i))
template <class C>
__global__ void __launch_bounds__(64) fb_build_kernel(int npairs, ModsetView ms, const uint32_t* __restrict__ h1,
                                                      const uint32_t* __restrict__ h2, uint32_t* __restrict__ tab) {
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
  uint32_t one[C::L], base[C::L], cur[C::L];
  load_owner<C>(one, ms.one_limbs + (size_t)st * C::K, ln);
  load_words_as_limbs<C>(cur, gl, hw, 0, ln);
  // step -1: base = Mont(h).  Then per window: 14 products e_d = e_{d-1} * base (d = 2..15), 4 squarings of base.
  const int nsteps = FB_MAX_WINDOWS * 18;
#pragma unroll 1
  for (int s = -1; s < nsteps; ++s) {
    const int i = s < 0 ? 0 : s / 18, ph = s < 0 ? -1 : s % 18;
    uint32_t* row = tab + ((size_t)pair * FB_MAX_WINDOWS + i) * 16 * C::K;
    if (s < 0) {
      copy_to_lds<C>(gl, ms.r2_limbs + (size_t)st * C::K, ln);
    } else if (ph == 0) {
      // new window: entries 0 and 1, multiplier = base, running product starts at base
      if (active) { store_owner<C>(row, one, ln); store_owner<C>(row + C::K, base, ln); }
      put_limbs<C>(gl, base, ln);
#pragma unroll
      for (int k = 0; k < C::L; ++k) cur[k] = base[k];
    } else if (ph == 14) {
      put_limbs<C>(gl, base, ln);                    // squarings: base <- base^2, four times
#pragma unroll
      for (int k = 0; k < C::L; ++k) cur[k] = base[k];
    } else if (ph > 14) {
      put_limbs<C>(gl, cur, ln);
    }
    wave_lds_sync();
    uint32_t r[C::L];
    montmul<C>(r, cur, gl, n, n0inv, ln);
    wave_lds_sync();
    reduce_once<C>(r, n, ln);                        // keep table entries canonical
#pragma unroll
    for (int k = 0; k < C::L; ++k) cur[k] = r[k];
    if (s < 0) {
#pragma unroll
      for (int k = 0; k < C::L; ++k) base[k] = cur[k];
    } else if (ph < 14) {
      if (active) store_owner<C>(row + (size_t)(ph + 2) * C::K, cur, ln);
    } else if (ph == 17) {
#pragma unroll
      for (int k = 0; k < C::L; ++k) base[k] = cur[k];
    }
  }
}

// out[i] = h^exp[i] mod N~ for h = h1 (which = 0) or h2 (which = 1) of statement st(i)
template <class C>
__global__ void __launch_bounds__(64) fb_modexp_kernel(int batch, ModsetView ms, Rows st_sel, int which,
                                                       const uint32_t* __restrict__ tab, Rows exps, int exp_words,
                                                       uint32_t* __restrict__ out) {
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const int nslots = gridDim.x * C::GROUPS;
  const int trips = (batch + nslots - 1) / nslots;
  const int nwin = exp_words * 8;
#pragma unroll 1
  for (int trip = 0; trip < trips; ++trip) {
    const int inst = trip * nslots + slot;
    const bool active = inst < batch;
    const int idx = active ? inst : batch - 1;
    const int st = sel_index(st_sel, idx);
    const uint32_t* ex = row_of(exps, idx);
    const uint32_t* T = tab + (size_t)(2 * st + which) * FB_MAX_WINDOWS * 16 * C::K;
    uint32_t n[C::L];
    load_owner<C>(n, ms.n_limbs + (size_t)st * C::K, ln);
    const uint32_t n0inv = ms.n0inv[st];
    uint32_t cur[C::L], nx[C::L];
    {
      const uint32_t d0 = ex[0] & 15u;
      load_owner<C>(cur, T + (size_t)d0 * C::K, ln);
    }
    // steps 1..nwin-1: cur <- cur * T[i][digit_i]; step nwin: cur <- cur * 1.  The next table row is fetched
    // (coalesced within the group) before the multiplication that hides its latency.
    auto fetch = [&](int i) {
      const uint32_t d = (ex[i >> 3] >> ((i & 7) * 4)) & 15u;
      const uint32_t* src = T + ((size_t)i * 16 + d) * C::K;
#pragma unroll
      for (int k = 0; k < C::L; ++k) nx[k] = src[ln.t + C::TPI * k];
    };
    if (nwin > 1) fetch(1);
#pragma unroll 1
    for (int s = 1; s <= nwin; ++s) {
      if (s < nwin) {
#pragma unroll
        for (int k = 0; k < C::L; ++k) gl[ln.t + C::TPI * k] = nx[k];
        if (s + 1 < nwin) fetch(s + 1);
      } else {
#pragma unroll
        for (int k = 0; k < C::L; ++k) gl[ln.t * C::L + k] = (ln.t == 0 && k == 0) ? 1u : 0u;
      }
      wave_lds_sync();
      uint32_t r[C::L];
      montmul<C>(r, cur, gl, n, n0inv, ln);
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < C::L; ++k) cur[k] = r[k];
    }
    reduce_once<C>(cur, n, ln);
    store_limbs_as_words<C>(out + (size_t)idx * C::K32, gl, cur, active, ln);
  }
}

}  // namespace mpe
