// Heavy kernels: modulus set-up, batched modular exponentiation and multiplication.
// (device code only; the C-ABI wrappers live in mpe_lib.hip)
#pragma once
#include "mpe_bigint.h"
#include "mpe_sched.h"

namespace mpe {

// A batch operand: row(i) = p + (idx ? idx[j] : j) * stride words with j = i >> shift.  stride = 0 broadcasts
// row 0.  words = number of valid words in a row (the rest of the integer is zero); 0 means "full width".
// shift = 1 lets the two CRT halves 2i, 2i+1 of item i read the same operand row.
struct Rows {
  const uint32_t* p;
  const int32_t* idx;
  int stride;
  int words;
  int shift;
};
__device__ __forceinline__ const uint32_t* row_of(const Rows& r, int i) {
  const int j = i >> r.shift;
  return r.p + (size_t)(r.idx ? r.idx[j] : j) * (size_t)r.stride;
}
// a selector (p unused): idx != null -> idx[j];  stride != 0 -> j;  else 0
__device__ __forceinline__ int sel_index(const Rows& sel, int i) {
  const int j = i >> sel.shift;
  return sel.idx ? sel.idx[j] : (sel.stride ? j : 0);
}

struct ModsetView {
  const uint32_t* n_limbs;
  const uint32_t* one_limbs;
  const uint32_t* r2_limbs;
  const uint32_t* r2h_limbs;   // 2^BITS * R^2 mod n (for double-width bases)
  const uint32_t* n0inv;
  int count;
};

// ---------------------------------------------------------------------------------------------
// modulus set-up kernel: one group per modulus
//   n limbs, n0inv = -n^-1 mod 2^W, one = R mod n, r2 = R^2 mod n, r2h = 2^BITS R^2 mod n
//   (R = 2^(W*K))
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) modset_setup_kernel(int count, const uint32_t* __restrict__ moduli,
                                                          uint32_t* __restrict__ n_limbs,
                                                          uint32_t* __restrict__ one_limbs,
                                                          uint32_t* __restrict__ r2_limbs,
                                                          uint32_t* __restrict__ r2h_limbs,
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
  // Mont(2^e) by square-and-double in the Montgomery domain, for e = W*K (-> r2) and e = BITS
  // (-> Mont(2^BITS), then r2h = montmul(r2, Mont(2^BITS)) = 2^BITS R^2).  One montmul call site:
  // pass 0 computes r2, pass 1 computes Mont(2^BITS), pass 2 is the single product for r2h.
  uint32_t mont2[C::L], r2[C::L], y[C::L];
#pragma unroll
  for (int i = 0; i < C::L; ++i) mont2[i] = one[i];
  dbl(mont2);                                         // Mont(2^1)
#pragma unroll
  for (int i = 0; i < C::L; ++i) { y[i] = mont2[i]; r2[i] = 0; }
  constexpr int E0 = C::W * C::K, E1 = C::BITS;
  constexpr int TOP0 = 31 - __builtin_clz((unsigned)E0), TOP1 = 31 - __builtin_clz((unsigned)E1);
  int pass = 0, b = TOP0 - 1;
#pragma unroll 1
  while (pass < 3) {
    if (pass == 2) {
      put_limbs<C>(gl, r2, ln);                      // y = Mont(2^BITS); multiply by r2
    } else {
      put_limbs<C>(gl, y, ln);                       // square
    }
    wave_lds_sync();
    uint32_t r[C::L];
    montmul<C>(r, y, gl, n, n0inv, ln);
    wave_lds_sync();
    reduce_once<C>(r, n, ln);
#pragma unroll
    for (int i = 0; i < C::L; ++i) y[i] = r[i];
    if (pass == 2) break;
    const int E = pass == 0 ? E0 : E1;
    if ((E >> b) & 1) dbl(y);
    if (--b < 0) {
      if (pass == 0) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) { r2[i] = y[i]; y[i] = mont2[i]; }
        b = TOP1 - 1;
      }
      ++pass;
    }
  }
  if (active) {
#pragma unroll
    for (int i = 0; i < C::L; ++i) {
      const size_t o = (size_t)idx * C::K + ln.t * C::L + i;
      n_limbs[o] = n[i];
      one_limbs[o] = one[i];
      r2_limbs[o] = r2[i];
      r2h_limbs[o] = y[i];
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
// interface words in global memory -> this lane's limbs (through the group's LDS region)
template <class C>
__device__ __forceinline__ void load_words_as_limbs(uint32_t (&v)[C::L], uint32_t* gl, const uint32_t* __restrict__ src,
                                                    int nwords, const Lane& ln) {
  stage_words<C>(gl, src, ln, nwords ? nwords : C::K32);
  wave_lds_sync();
  limbs_from_words<C>(v, gl, ln);
  wave_lds_sync();
}
// canonical residue (exact limbs) -> interface words in global memory
template <class C>
__device__ __forceinline__ void store_limbs_as_words(uint32_t* __restrict__ dst, uint32_t* gl, const uint32_t (&v)[C::L],
                                                     bool active, const Lane& ln) {
  put_limbs<C>(gl, v, ln);
  if (ln.t0) { gl[C::K] = 0; gl[C::K + 1] = 0; }
  wave_lds_sync();
  if (active)
    for (int q = ln.t; q < C::K32; q += C::TPI) dst[q] = word_from_limbs<C>(gl, q);
  wave_lds_sync();
}

// window wi (wb bits) of a little-endian exponent; windows may straddle words, the top one may be partial
__device__ __forceinline__ uint32_t exp_window(const uint32_t* __restrict__ ex, int exp_words, int wi, int wb) {
  const int bitpos = wi * wb, word = bitpos >> 5, sh = bitpos & 31;
  uint32_t v = ex[word] >> sh;
  if (sh + wb > 32 && word + 1 < exp_words) v |= ex[word + 1] << (32 - sh);
  return v & ((1u << wb) - 1u);
}

// ---------------------------------------------------------------------------------------------
// modexp kernel: persistent waves, each group walks the batch with a grid stride.
//   out[i] = (base_lo[i] + 2^BITS * base_hi[i]) ^ exp[i]  [ * base2[i] ^ exp2[i] ]   mod  modulus[mod(i)]
// base_hi.p == nullptr -> single-width base.  base2.p != nullptr -> the product of two powers on one ladder
// (the squarings are shared; exp2 is short: 4-bit windows, 32 exp2_words <= (nwin-1) wb).
// Fixed windows, constant operation sequence: it depends on the exponent LENGTHS only.
// ---------------------------------------------------------------------------------------------
enum ModexpPhase { PH_WIDE, PH_MONT1, PH_TAB1, PH_MONT2, PH_TAB2, PH_SQ, PH_MUL1, PH_MUL2, PH_FINAL, PH_DONE };

template <class C>
__global__ void __launch_bounds__(64) modexp_kernel(int batch, ModsetView ms, Rows mod_sel, Rows base_lo, Rows base_hi,
                                                    Rows exps, int exp_words, int wb, Rows base2, Rows exps2,
                                                    int exp2_words, uint32_t* __restrict__ out,
                                                    uint32_t* __restrict__ tables, SchedArgs sched) {
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const int nslots = gridDim.x * C::GROUPS;
  const bool wide = base_hi.p != nullptr, dual = base2.p != nullptr;
  const int TE = 1 << wb;                                // window width wb in {4,5,6}: 2^wb table entries
  // this group's window tables: tab[0..TE-1] = Mont(base^d); tab2[1..15] = Mont(base2^d) (digit 0 uses tab[0] = Mont(1))
  uint32_t* tab = tables + (size_t)slot * (TE + (dual ? 16 : 0)) * C::K;
  uint32_t* tab2 = tab + (size_t)TE * C::K;
  const int nwin = (exp_words * 32 + wb - 1) / wb;
  const int nwin2 = dual ? exp2_words * 8 : 0;
  const int top_bit = (nwin - 1) * wb;                   // the ladder squares top_bit times
  // One Montgomery multiplication per step; the phase (wave-uniform) decides where the multiplier comes from
  // and where the product goes, so montmul is instantiated exactly once:
  //   PH_WIDE            : hi * (2^BITS R^2)           -> Mont(2^BITS hi), kept aside (in the table slab)
  //   PH_MONT1           : cur = base * R^2 (+ aside)  -> Mont(base) = tab[1]
  //   PH_TAB1 k=1..TE-2  : cur = cur * Mont(base)      -> tab[k+1]
  //   PH_MONT2 / PH_TAB2 : the same for base2          -> tab2[1..15]
  //   ladder             : PH_SQ per bit; PH_MUL1 by tab[window] where a window of exp starts, PH_MUL2 by
  //                        tab2[window] where a 4-bit window of exp2 starts
  //   PH_FINAL           : cur = cur * 1               -> leaves the Montgomery domain

  WaveSched ws;                                          // mpe_sched.h: which unit this wave runs next
  ws.init(sched);
#pragma unroll 1
  for (;;) {
    int ubase;
    if (!ws.next(sched, batch, nslots, C::GROUPS, ubase)) break;           // nothing left for this wave (wave-uniform)
    const int inst = ubase + ln.g;
    const bool active = inst < batch;
    const int idx = active ? inst : batch - 1;
    const int mi = sel_index(mod_sel, idx);
    const uint32_t* ex = row_of(exps, idx);
    const uint32_t* ex2 = dual ? row_of(exps2, idx) : ex;

    uint32_t n[C::L];
    load_owner<C>(n, ms.n_limbs + (size_t)mi * C::K, ln);
    const uint32_t n0inv = ms.n0inv[mi];

    uint32_t cur[C::L];
    load_words_as_limbs<C>(cur, gl, wide ? row_of(base_hi, idx) : row_of(base_lo, idx),
                           wide ? base_hi.words : base_lo.words, ln);
    {
      uint32_t one[C::L];
      load_owner<C>(one, ms.one_limbs + (size_t)mi * C::K, ln);
      store_owner<C>(tab, one, ln);                         // tab[0] = Mont(1)
    }

    int ph = wide ? PH_WIDE : PH_MONT1;
    int k = 0;                                              // position inside a table phase
    int b = top_bit;                                        // the ladder has consumed the exponent bits >= b
#pragma unroll 1
    while (ph != PH_DONE) {
      // ---- multiplier -> LDS ----
      if (ph == PH_WIDE) {
        copy_to_lds<C>(gl, ms.r2h_limbs + (size_t)mi * C::K, ln);
      } else if (ph == PH_MONT1 || ph == PH_MONT2) {
        copy_to_lds<C>(gl, ms.r2_limbs + (size_t)mi * C::K, ln);
      } else if (ph == PH_TAB1 || ph == PH_TAB2) {
        if (k == 1) put_limbs<C>(gl, cur, ln);              // Mont(base) stays in LDS for the whole table phase
      } else if (ph == PH_SQ) {
        put_limbs<C>(gl, cur, ln);
      } else if (ph == PH_MUL1) {
        const uint32_t w = exp_window(ex, exp_words, b / wb, wb);
        copy_to_lds<C>(gl, tab + (size_t)w * C::K, ln);
      } else if (ph == PH_MUL2) {
        const int wi = b >> 2;
        const uint32_t w = (ex2[wi >> 3] >> ((wi & 7) * 4)) & 15u;
        copy_to_lds<C>(gl, w ? tab2 + (size_t)w * C::K : tab, ln);
      } else {                                              // PH_FINAL
#pragma unroll
        for (int i = 0; i < C::L; ++i) gl[ln.t * C::L + i] = (ln.t == 0 && i == 0) ? 1u : 0u;
      }
      wave_lds_sync();
      uint32_t r[C::L];
      montmul<C>(r, cur, gl, n, n0inv, ln);
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < C::L; ++i) cur[i] = r[i];

      // ---- product -> its place; next phase ----
      bool tables_done = false;
      if (ph == PH_WIDE) {
        store_owner<C>(tab + 2 * C::K, cur, ln);            // kept aside in the (still empty) table slab, not in registers
        load_words_as_limbs<C>(cur, gl, row_of(base_lo, idx), base_lo.words, ln);
        ph = PH_MONT1;
      } else if (ph == PH_MONT1) {
        if (wide) {
          // Mont(lo) + Mont(2^BITS hi): value < 4n, settle the limbs exactly before it is squared
          uint32_t aside[C::L];
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          load_owner<C>(aside, tab + 2 * C::K, ln);
          int64_t z[C::L];
#pragma unroll
          for (int i = 0; i < C::L; ++i) z[i] = (int64_t)cur[i] + (int64_t)aside[i];
          full_normalize<C>(z, ln);
#pragma unroll
          for (int i = 0; i < C::L; ++i) cur[i] = (uint32_t)z[i];
        }
        store_owner<C>(tab + C::K, cur, ln);
        k = 1;
        ph = PH_TAB1;                                       // TE >= 16: at least tab[2..] to fill
      } else if (ph == PH_TAB1) {
        store_owner<C>(tab + (size_t)(k + 1) * C::K, cur, ln);
        if (++k > TE - 2) {
          if (dual) {
            load_words_as_limbs<C>(cur, gl, row_of(base2, idx), base2.words, ln);
            ph = PH_MONT2;
          } else {
            tables_done = true;
          }
        }
      } else if (ph == PH_MONT2) {
        store_owner<C>(tab2 + C::K, cur, ln);
        k = 1;
        ph = PH_TAB2;
      } else if (ph == PH_TAB2) {
        store_owner<C>(tab2 + (size_t)(k + 1) * C::K, cur, ln);
        if (++k > 14) tables_done = true;
      } else if (ph == PH_FINAL) {
        ph = PH_DONE;
      } else {
        // ladder: after a squaring the position moves down one bit; multiplications leave it in place
        if (ph == PH_SQ) --b;
        const bool m1 = ph == PH_SQ && (b % wb) == 0;
        const bool m2 = ph != PH_MUL2 && dual && (b & 3) == 0 && (b >> 2) < nwin2;
        ph = m1 ? PH_MUL1 : (m2 ? PH_MUL2 : (b == 0 ? PH_FINAL : PH_SQ));
      }
      if (tables_done) {
        // tables complete: start the ladder from the top window of the long exponent
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const uint32_t w = exp_window(ex, exp_words, nwin - 1, wb);
        load_owner<C>(cur, tab + (size_t)w * C::K, ln);
        b = top_bit;
        ph = b == 0 ? PH_FINAL : PH_SQ;
      }
    }
    // canonical residue -> interface words
    reduce_once<C>(cur, n, ln);
    store_limbs_as_words<C>(out + (size_t)idx * C::K32, gl, cur, active, ln);
  }
}

// ---------------------------------------------------------------------------------------------
// modmul kernel: out = a*b mod n   (two Montgomery multiplications: by b, then by R^2)
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) modmul_kernel(int batch, ModsetView ms, Rows mod_sel, Rows A, Rows B,
                                                    uint32_t* __restrict__ out) {
  MPE_FOREGROUND();
  __shared__ uint32_t lds[C::LDS_WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const int nslots = gridDim.x * C::GROUPS;
  const int trips = (batch + nslots - 1) / nslots;
#pragma unroll 1
  for (int trip = 0; trip < trips; ++trip) {
    if (trip * nslots + (int)blockIdx.x * C::GROUPS >= batch) break;       // no item left for this wave (wave-uniform)
    const int inst = trip * nslots + slot;
    const bool active = inst < batch;
    const int idx = active ? inst : batch - 1;
    const int mi = sel_index(mod_sel, idx);
    uint32_t n[C::L];
    load_owner<C>(n, ms.n_limbs + (size_t)mi * C::K, ln);
    const uint32_t n0inv = ms.n0inv[mi];

    uint32_t cur[C::L], b[C::L];
    load_words_as_limbs<C>(cur, gl, row_of(A, idx), A.words, ln);
    load_words_as_limbs<C>(b, gl, row_of(B, idx), B.words, ln);
#pragma unroll 1
    for (int step = 0; step < 2; ++step) {
      if (step == 0) put_limbs<C>(gl, b, ln);                                  // a*b/R
      else copy_to_lds<C>(gl, ms.r2_limbs + (size_t)mi * C::K, ln);           // * R^2 / R
      wave_lds_sync();
      uint32_t r[C::L];
      montmul<C>(r, cur, gl, n, n0inv, ln);
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < C::L; ++i) cur[i] = r[i];
    }
    reduce_once<C>(cur, n, ln);
    store_limbs_as_words<C>(out + (size_t)idx * C::K32, gl, cur, active, ln);
  }
}

}  // namespace mpe
