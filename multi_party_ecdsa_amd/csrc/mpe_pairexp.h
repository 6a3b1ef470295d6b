// Exponentiation modulo a SQUARE (N^2 for Paillier, p^2 | q^2 for the key holder's CRT halves) in N-adic form.
//
// Every 4096-bit exponentiation on the GG20 path is modulo N^2 (src/utilities/mta/range_proofs.rs:53-55,134-141,
// zk_pdl_with_slack/mod.rs:87-93,144-157, mta/mod.rs:68-75,133-145).  An element of Z/N^2 is kept as a pair
// (x0, x1) of residues mod N with   x0 + x1 N = v R  (mod N^2),   R = 2^(W K) the Montgomery radix of N.  Then
//
//     (x0 + x1 N)(y0 + y1 N) R^-1 = u + ((x0 y1 + x1 y0 - m) R^-1 mod N) N      (mod N^2)
//
// where x0 y0 + m N = u R is the ordinary Montgomery step mod N (u = its result, m = its quotient digits) —
// the x1 y1 N^2 term vanishes and every reduction is HALF-size.  A multiplication mod N^2 is therefore
//   pass A : u  = CIOS(x0, y0)  mod N, remembering the K quotient digits m_j
//   pass B : z1 = CIOS-reduce( x0 y1 + x1 y0 + [K_c + (R-1-m)] ),   K_c = -(R-1) mod N  (so the bracket = -m mod N,
//            limb-wise non-negative: it just pre-loads the accumulator columns)
// i.e. 5 K^2 multiply-accumulates, and a squaring (x0 x1 counted twice: one stream with 2 x1) 4 K^2 — against
// 8 K^2 for the Montgomery multiplication of the 2K-limb integers.  The residues produced are the same numbers
// mpz_powm returns: the pair is converted back (multiply by (1, 0), normalise, z0 + z1 N) at the end.
#pragma once
#include <type_traits>
#include "mpe_internal.h"
#include "mpe_small.h"

namespace mpe {

template <class C>
struct PairLds {
  static constexpr int B0 = 0;            // multiplier y0: K limbs
  static constexpr int B1 = C::K;         // multiplier y1: K limbs
  static constexpr int M = 2 * C::K;      // quotient digits of pass A: K limbs
  static constexpr int KC = 3 * C::K;     // K_c + (R - 1) of this item's modulus, limb-wise: K limbs
  // words per group: even (64-bit LDS reads stay aligned) and such that the groups of a 32-lane half hit distinct banks
  static constexpr int pick_stride() {
    constexpr int per_half = (C::GROUPS >= 2) ? C::GROUPS / 2 : 1;
    for (int s = 4 * C::K + 2;; s += 2) {
      bool ok = true;
      for (int a = 0; a < per_half && ok; ++a)
        for (int b = a + 1; b < per_half; ++b)
          if (((a * s) & 31) == ((b * s) & 31)) { ok = false; break; }
      if (ok) return s;
    }
  }
  static constexpr int STRIDE = pick_stride();
  static constexpr int WORDS = STRIDE * C::GROUPS;
};

// Limbs per component in the STORED per-modulus constants: the widest layout of that modulus size (16 lanes x 5 limbs = 80 for
// 2048 bit, 8 x 5 = 40 for 1024 bit), zero-padded — every layout (18 / 9 / 5 limbs per lane) reads the same arrays, its own K
// limbs of each component.  (R = 2^(W STEPS) does not depend on the layout, so neither do the constants.)
__host__ __device__ constexpr int pair_kstore(int bits) { return bits == 2048 ? 80 : (bits == 1024 ? 40 : 0); }

// per-modulus constants of the pair arithmetic (limb arrays, modulus-major, pair_kstore limbs per component)
struct PairsetView {
  const uint32_t* n_limbs;   // [count][K]
  const uint32_t* n0inv;     // [count]
  const uint32_t* one;       // [count][2K]  pair = R      (the form of 1)
  const uint32_t* r2;        // [count][2K]  pair = R^2    (multiplying a plain pair by it gives its form)
  const uint32_t* tp;        // [count][2K]  pair = 2^BITS R (the form of 2^BITS: Horner step over BITS-bit chunks)
  const uint32_t* kc;        // [count][K]   -(R-1) mod N
  int count;
};

// the shared tail of a CIOS pass: local ripple, then hand the lane's carry-out to the next lane without rippling on
template <class C>
__device__ __forceinline__ void cios_tail(uint32_t (&res)[C::L], const uint64_t (&c)[C::L], const Lane& ln) {
  constexpr int L = C::L, W = C::W;
  uint64_t carry = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const uint64_t v = c[i] + carry;
    res[i] = (uint32_t)v & C::MASK;
    carry = v >> W;
  }
  uint64_t cin = pull_prev64(carry);
  if (ln.t0) cin = 0;
  const uint32_t v0 = res[0] + ((uint32_t)cin & C::MASK);
  res[0] = v0 & C::MASK;                       // one extra ripple step: every limb < 2^W + 2^12
  res[1] += (uint32_t)(cin >> W) + (v0 >> W);
}

// The outer loop ends after STEPS = LAST_TRIP * L + REM steps (STEPS <= K: the multiplier has no limbs beyond STEPS): the rotating
// column names are then REM positions into a turn — logical column i sits in c[(REM + i) % L].  (A renaming: no instruction.)
template <class C>
__device__ __forceinline__ void cios_finish(uint32_t (&res)[C::L], const uint64_t (&c)[C::L], const Lane& ln) {
  static_assert(C::STEPS <= C::K && C::STEPS > C::K - 2 * C::L, "STEPS must lie in the last two trips");
  constexpr int REM = C::STEPS % C::L;
  if constexpr (REM == 0) {
    cios_tail<C>(res, c, ln);
  } else {
    uint64_t cc[C::L];
#pragma unroll
    for (int i = 0; i < C::L; ++i) cc[i] = c[(REM + i) % C::L];
    cios_tail<C>(res, cc, ln);
  }
}

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N) — the CIOS steps name their registers by the step index
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// One CIOS pass with ONE product stream:  res = (c_in + a * b + m n) / R,  quotient digits m_j stored to ml[j].
// (montmul of mpe_bigint.h with pre-loaded columns and the digits kept.)
// The loop has ONE exit, after STEPS = (STEPS / L) * L + STEPS % L steps (71 of 72 for 2048-bit moduli at 18 or 9 limbs per lane,
// 71 of 80 at 5 limbs per lane; all 36 for 1024 bit): the column names have one rotation state there (cios_finish), nothing for
// the compiler to reconcile.
template <class C, bool STORE_M>
__device__ __forceinline__ void cios1(uint32_t (&res)[C::L], uint64_t (&c)[C::L], const uint32_t (&a)[C::L],
                                      const uint32_t* __restrict__ bl, uint32_t* __restrict__ ml,
                                      const uint32_t (&n)[C::L], uint32_t n0inv, const Lane& ln) {
  constexpr int L = C::L, W = C::W;
  uint32_t maskv = C::MASK;
  asm volatile("" : "+v"(maskv));
#pragma unroll 1
  for (int jj = 0;; ++jj) {
    const uint32_t* bp = bl + jj * L;
    uint32_t* mp = ml + jj * L;
    auto step = [&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const uint32_t bj = bp[r];
      c[r] += (uint64_t)a[0] * bj;
      const uint32_t m = bcast0_masked<C::TPI>((uint32_t)c[r] * n0inv, maskv);
      if (STORE_M) mp[r] = m;                    // every lane of the group writes the same word
#pragma unroll
      for (int i = 1; i < L; ++i) c[(r + i) % L] += (uint64_t)a[i] * bj;
#pragma unroll
      for (int i = 0; i < L; ++i) c[(r + i) % L] += (uint64_t)m * n[i];
      c[(r + 1) % L] += c[r] >> W;
      c[r] = (uint64_t)(pull_next((uint32_t)c[r]) & maskv);
    };
    static_for<0, C::STEPS % L>(step);
    if (jj == C::STEPS / L) break;                            // the ONLY exit: after STEPS steps (R = 2^(W STEPS))
    static_for<C::STEPS % L, L>(step);
  }
  cios_finish<C>(res, c, ln);
}

// One CIOS pass with TWO product streams:  res = (c_in + a0 * b1 + a1 * b0 + m n) / R
template <class C>
__device__ __forceinline__ void cios2(uint32_t (&res)[C::L], uint64_t (&c)[C::L], const uint32_t (&a0)[C::L],
                                      const uint32_t (&a1)[C::L], const uint32_t* __restrict__ bl0,
                                      const uint32_t* __restrict__ bl1, const uint32_t (&n)[C::L], uint32_t n0inv,
                                      const Lane& ln) {
  constexpr int L = C::L, W = C::W;
  uint32_t maskv = C::MASK;
  asm volatile("" : "+v"(maskv));
#pragma unroll 1
  for (int jj = 0;; ++jj) {
    const uint32_t* bp0 = bl0 + jj * L;
    const uint32_t* bp1 = bl1 + jj * L;
    auto step = [&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const uint32_t b0j = bp0[r], b1j = bp1[r];
      c[r] += (uint64_t)a0[0] * b1j;
      c[r] += (uint64_t)a1[0] * b0j;
      const uint32_t m = bcast0_masked<C::TPI>((uint32_t)c[r] * n0inv, maskv);
#pragma unroll
      for (int i = 1; i < L; ++i) {
        c[(r + i) % L] += (uint64_t)a0[i] * b1j;
        c[(r + i) % L] += (uint64_t)a1[i] * b0j;
      }
#pragma unroll
      for (int i = 0; i < L; ++i) c[(r + i) % L] += (uint64_t)m * n[i];
      c[(r + 1) % L] += c[r] >> W;
      c[r] = (uint64_t)(pull_next((uint32_t)c[r]) & maskv);
    };
    static_for<0, C::STEPS % L>(step);
    if (jj == C::STEPS / L) break;                            // the ONLY exit: after STEPS steps (R = 2^(W STEPS))
    static_for<C::STEPS % L, L>(step);
  }
  cios_finish<C>(res, c, ln);
}

// (r0, r1) = (a0, a1) * (y0, y1) R^-1 in Z/N^2.  The group's LDS region holds y0 in B0 and y1 in B1; for a squaring
// (sq: y == a) B1 holds 2 y0 instead, so that pass B is the single stream a1 * (2 a0).
// Column bound: a pass-B column absorbs per lane block 18 x (2^59.01 + 2^58.01) (squaring: the doubled stream) or
// 18 x 3 x 2^58.01 (two streams) plus the 2^30 pre-load and the fold carries: < 2^63.8.
template <class C>
__device__ __forceinline__ void pairmul(uint32_t (&r0)[C::L], uint32_t (&r1)[C::L], const uint32_t (&a0)[C::L],
                                        const uint32_t (&a1)[C::L], uint32_t* gl, const uint32_t (&n)[C::L],
                                        uint32_t n0inv, bool sq, bool half, const Lane& ln) {
  using PL = PairLds<C>;
  constexpr int L = C::L;
  uint64_t c[L];
#pragma unroll
  for (int i = 0; i < L; ++i) c[i] = 0;
  cios1<C, true>(r0, c, a0, gl + PL::B0, gl + PL::M, n, n0inv, ln);           // pass A: u, digits -> M
  wave_lds_sync();
  if (half) {                                      // arithmetic modulo N only: the x1 components stay 0
#pragma unroll
    for (int i = 0; i < L; ++i) r1[i] = 0;
    return;
  }
#pragma unroll
  for (int i = 0; i < L; ++i) {
    uint32_t mi = gl[PL::M + ln.t * L + i];
    if (C::STEPS == C::K - 1) {
      if (i == L - 1) mi = (ln.t == C::TPI - 1) ? 0u : mi;                       // there is no digit m_(K-1)
    } else if (C::STEPS < C::K) {
      mi = (ln.t * L + i < C::STEPS) ? mi : 0u;                                  // nor any digit beyond STEPS
    }
    c[i] = (uint64_t)(gl[PL::KC + ln.t * L + i] - mi);
  }
  wave_lds_sync();
  // u waits in the M region (its digits are consumed) so that pass B does not carry 18 more live registers
#pragma unroll
  for (int i = 0; i < L; ++i) gl[PL::M + ln.t * L + i] = r0[i];
  if (sq) {
    cios1<C, false>(r1, c, a1, gl + PL::B1, gl + PL::M, n, n0inv, ln);        // pass B: a1 * (2 a0) - m
  } else {
    cios2<C>(r1, c, a0, a1, gl + PL::B0, gl + PL::B1, n, n0inv, ln);          // pass B: a0 * y1 + a1 * y0 - m
  }
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < L; ++i) r0[i] = gl[PL::M + ln.t * L + i];
}

// off1: where the second component starts in `src` (C::K in a window table, pair_kstore in the per-modulus constants)
template <class C>
__device__ __forceinline__ void copy_pair_to_lds(uint32_t* gl, const uint32_t* __restrict__ src, const Lane& ln, int off1 = C::K) {
  using PL = PairLds<C>;
#pragma unroll
  for (int i = 0; i < C::L; ++i) {
    gl[PL::B0 + ln.t + C::TPI * i] = src[ln.t + C::TPI * i];
    gl[PL::B1 + ln.t + C::TPI * i] = src[off1 + ln.t + C::TPI * i];
  }
}
// A window-table row is y0 | y1 = 2K CONTIGUOUS words, and so are B0 | B1 in the group's LDS region: the multiplier of a
// window multiplication moves as 16-byte global loads (a quarter of the memory instructions of the word-by-word copy, one
// request per 4 limbs) and 8-byte LDS stores (the group regions are 8-byte aligned: PairLds::STRIDE is even).
template <class C>
__device__ __forceinline__ void copy_row_to_lds(uint32_t* gl, const uint32_t* __restrict__ row, const Lane& ln) {
  using PL = PairLds<C>;
  static_assert(PL::B0 == 0 && PL::B1 == C::K && (2 * C::K) % 4 == 0 && PL::STRIDE % 2 == 0, "B0 | B1 must be one 8-byte aligned run");
  constexpr int CHUNKS = 2 * C::K / 4;                       // 16-byte chunks of the row
  const uint4* __restrict__ r4 = reinterpret_cast<const uint4*>(row);
  uint2* l2 = reinterpret_cast<uint2*>(gl + PL::B0);
#pragma unroll
  for (int i = 0; i < (CHUNKS + C::TPI - 1) / C::TPI; ++i) {
    const int ch = ln.t + C::TPI * i;
    if ((i + 1) * C::TPI <= CHUNKS || ch < CHUNKS) {
      const uint4 v = r4[ch];
      l2[2 * ch] = make_uint2(v.x, v.y);
      l2[2 * ch + 1] = make_uint2(v.z, v.w);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// constants: one group per modulus.  Everything comes from exact doublings of the pair (1, 0):
//   after W K doublings the pair is R (= `one`), after W K + BITS it is 2^BITS R (= `tp`), after 2 W K it is R^2.
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) pairset_setup_kernel(int count, const uint32_t* __restrict__ moduli,
                                                           uint32_t* __restrict__ n_limbs, uint32_t* __restrict__ n0inv_out,
                                                           uint32_t* __restrict__ one, uint32_t* __restrict__ r2,
                                                           uint32_t* __restrict__ tp, uint32_t* __restrict__ kc) {
  __shared__ uint32_t lds[C::LDS_WORDS];
  constexpr int KS = pair_kstore(C::BITS);          // storage stride (the arrays were zeroed: the padding limbs stay 0)
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * C::STRIDE;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const bool active = slot < count;
  const int idx = active ? slot : count - 1;
  stage_words<C>(gl, moduli + (size_t)idx * C::K32, ln);
  wave_lds_sync();
  uint32_t n[C::L];
  limbs_from_words<C>(n, gl, ln);
  const uint32_t w0 = gl[0];
  uint32_t inv = w0;
#pragma unroll
  for (int i = 0; i < 5; ++i) inv *= 2u - w0 * inv;
  const uint32_t n0inv = (0u - inv) & C::MASK;
  wave_lds_sync();

  int64_t x0[C::L], x1[C::L];
#pragma unroll
  for (int i = 0; i < C::L; ++i) { x0[i] = (ln.t == 0 && i == 0) ? 1 : 0; x1[i] = 0; }
  auto store_pair = [&](uint32_t* dst) {
    if (!active) return;
#pragma unroll
    for (int i = 0; i < C::L; ++i) {
      dst[(size_t)idx * 2 * KS + ln.t * C::L + i] = (uint32_t)x0[i];
      dst[(size_t)idx * 2 * KS + KS + ln.t * C::L + i] = (uint32_t)x1[i];
    }
  };
  constexpr int WK = C::W * C::STEPS;          // R = 2^WK: the radix the CIOS passes divide by
#pragma unroll 1
  for (int d = 1; d <= 2 * WK; ++d) {
    // (x0, x1) <- 2 (x0, x1):  x0 = 2 x0 [- N, carry 1];  x1 = 2 x1 + carry [- N]
#pragma unroll
    for (int i = 0; i < C::L; ++i) x0[i] *= 2;
    full_normalize<C>(x0, ln);
    const bool cy = cmp_ge<C>(x0, n, ln);
    if (cy) {
#pragma unroll
      for (int i = 0; i < C::L; ++i) x0[i] -= (int64_t)n[i];
      full_normalize<C>(x0, ln);
    }
#pragma unroll
    for (int i = 0; i < C::L; ++i) x1[i] *= 2;
    if (cy && ln.t == 0) x1[0] += 1;
    full_normalize<C>(x1, ln);
    if (cmp_ge<C>(x1, n, ln)) {
#pragma unroll
      for (int i = 0; i < C::L; ++i) x1[i] -= (int64_t)n[i];
      full_normalize<C>(x1, ln);
    }
    if (d == WK) {
      store_pair(one);
      // kc = -(R - 1) mod N = N + 1 - (R mod N), reduced
      int64_t z[C::L];
#pragma unroll
      for (int i = 0; i < C::L; ++i) z[i] = (int64_t)n[i] - x0[i] + ((ln.t == 0 && i == 0) ? 1 : 0);
      full_normalize<C>(z, ln);
      if (cmp_ge<C>(z, n, ln)) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) z[i] -= (int64_t)n[i];
        full_normalize<C>(z, ln);
      }
      if (active) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) kc[(size_t)idx * KS + ln.t * C::L + i] = (uint32_t)z[i];
      }
    }
    if (d == WK + C::BITS) store_pair(tp);
  }
  store_pair(r2);
  if (active) {
#pragma unroll
    for (int i = 0; i < C::L; ++i) n_limbs[(size_t)idx * KS + ln.t * C::L + i] = n[i];
    if (ln.t0) n0inv_out[idx] = n0inv;
  }
}

// ---------------------------------------------------------------------------------------------
// (half != 0: the same machine on the x0 components alone = plain Montgomery exponentiation modulo N; out = value < N)
// exponentiation modulo modulus[mod(i)]^2:   out pair (z0 | z1, each K32 words, both in [0, N)) with
//   z0 + z1 N = base^exp [* base2^exp2]  mod N^2        (pair_finish_kernel then forms that integer in place)
// base rows may be any number of words (chunks of K32 words, Horner); phases as in modexp_kernel.
// ---------------------------------------------------------------------------------------------
enum PairPhase { PP_IN, PP_MONT, PP_TSQ, PP_TAB, PP_SQ, PP_MUL1, PP_MUL2, PP_FINAL, PP_DONE };

// Sliding windows for a PUBLIC exponent that the whole wave shares (every exponentiation of the dominant kernel raises to the
// public key N: s^N of the range-proof verifiers, r^N of MessageB — range_proofs.rs:134-141, zk_pdl_with_slack/mod.rs:144-157,
// mta/mod.rs:133-137): the highest set bit at or below `from` starts a window of at most wb bits that ENDS in a set bit.
// Returns its low end (-1: no bit left), val = its (odd) value.  All operands are wave-uniform: scalar control flow.
// (ex lives in the constant address space: the key table is never written while the kernel runs, so the reads are scalar
// loads and the compiler can see that the schedule is wave-uniform — a generic pointer would make every load "divergent")
typedef const uint32_t __attribute__((address_space(4))) * UniformWords;
__device__ __forceinline__ int slide_window(UniformWords ex, int exp_words, int from, int wb, uint32_t& val) {
  if (from >= exp_words * 32) from = exp_words * 32 - 1;
  if (from < 0) return -1;
  int w = from >> 5;
  uint32_t x = ex[w] & (0xFFFFFFFFu >> (31 - (from & 31)));
  while (x == 0) {
    if (--w < 0) return -1;
    x = ex[w];
  }
  const int hi = w * 32 + 31 - __clz((int)x);
  int lo = hi - wb + 1;
  if (lo < 0) lo = 0;
  const int q = lo >> 5, sh = lo & 31;
  uint64_t two = ex[q];
  if (q + 1 < exp_words) two |= (uint64_t)ex[q + 1] << 32;
  const uint32_t v = (uint32_t)(two >> sh) & ((1u << (hi - lo + 1)) - 1u);
  const int tz = __ffs((int)v) - 1;                 // v != 0: bit hi is set
  val = v >> tz;
  return lo + tz;
}

// items ordered by key, so that the 16 exponentiations of a wave share their (public) exponent: LDS histogram per block,
// one global atomic per (block, key present), then the scatter.  The order inside a key is arbitrary (results do not depend on it).
static __global__ void __launch_bounds__(256) key_hist_kernel(int batch, Rows sel, int nkeys, int32_t* __restrict__ cnt) {
  extern __shared__ int32_t sh[];
  for (int k = threadIdx.x; k < nkeys; k += blockDim.x) sh[k] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) atomicAdd(&sh[sel_index(sel, i)], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < nkeys; k += blockDim.x)
    if (sh[k]) atomicAdd(&cnt[k], sh[k]);
}
static __global__ void __launch_bounds__(64) key_scan_kernel(int nkeys, const int32_t* __restrict__ cnt, int32_t* __restrict__ cursor) {
  // exclusive prefix sums by one wave: each lane sums a contiguous chunk, a DPP-free serial pass over the 64 partials
  __shared__ int32_t part[64];
  const int per = (nkeys + 63) / 64, lo = threadIdx.x * per, hi = lo + per < nkeys ? lo + per : nkeys;
  int32_t s = 0;
  for (int k = lo; k < hi; ++k) s += cnt[k];
  part[threadIdx.x] = s;
  __syncthreads();
  int32_t base = 0;
  for (int t = 0; t < (int)threadIdx.x; ++t) base += part[t];
  for (int k = lo; k < hi; ++k) { cursor[k] = base; base += cnt[k]; }
}
static __global__ void __launch_bounds__(256) key_scatter_kernel(int batch, Rows sel, int nkeys, int32_t* __restrict__ cursor, int32_t* __restrict__ perm) {
  extern __shared__ int32_t sh[];                   // [nkeys] counts of this block, [nkeys] their global bases
  for (int k = threadIdx.x; k < 2 * nkeys; k += blockDim.x) sh[k] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int mi = i < batch ? sel_index(sel, i) : -1;
  const int local = mi >= 0 ? atomicAdd(&sh[mi], 1) : 0;
  __syncthreads();
  for (int k = threadIdx.x; k < nkeys; k += blockDim.x)
    if (sh[k]) sh[nkeys + k] = atomicAdd(&cursor[k], sh[k]);
  __syncthreads();
  if (mi >= 0) perm[sh[nkeys + mi] + local] = i;
}

// The sliding decision of pair_modexp_kernel<C, true>, replayed per UNIT (the `groups` consecutive positions one wave carries through
// one ladder — whichever wave the scheduler gives it to) for the profiler (mpe_prof_rec.sliding_frac): a unit slides iff its
// exponentiations read the same exponent row and the first window stays above a second exponent.
static __global__ void pair_slide_audit_kernel(int batch, int groups, Rows exps, int exp_words, int wb, int exp2_words,
                                               const int32_t* __restrict__ perm, uint32_t* __restrict__ ctr) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u * groups >= batch) return;
  const uint32_t* first = nullptr;
  bool same = true;
  for (int g = 0; g < groups; ++g) {
    const int inst = u * groups + g;
    const int pos = inst < batch ? inst : batch - 1;
    const uint32_t* ex = row_of(exps, perm ? perm[pos] : pos);
    if (g == 0) first = ex; else same = same && ex == first;
  }
  if (!same) return;
  uint32_t val = 0;
  const int lo = slide_window((UniformWords)(uintptr_t)first, exp_words, exp_words * 32 - 1, wb, val);
  if (lo < 0 || (exp2_words && lo < 32 * exp2_words)) return;
  atomicAdd(ctr, 1u);
}

#ifdef MPE_WAVE_TRACE
// PROFILING BUILD ONLY (build.sh -DMPE_WAVE_TRACE -> a library of its own, loaded through MPE_LIB_PATH by tools/trace_waves.py; never
// shipped): every (wave, trip) of the ladder kernel leaves one record — where it ran (HW_ID, XCC_ID) and when (s_memtime = the
// shader-clock counter, s_memrealtime = the constant 100 MHz counter; their ratio is the clock the wave really saw).  Written to
// settle round 5's open question: why a FRESH launch of <= 1 024 ladder waves runs its trip at 0.95 of a shared trip while the
// TAIL of a full grid runs it at 0.49 — placement, clock, or issue arbitration.
static __device__ unsigned long long* g_wave_trace;
static __device__ unsigned g_wave_trace_cap, g_wave_trace_n;
#endif

template <class C, bool SLIDE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) pair_modexp_kernel(int batch, PairsetView ps, Rows mod_sel, Rows base, Rows exps,
                                                         int exp_words, int wb, Rows base2, Rows exps2, int exp2_words,
                                                         int half, uint32_t* __restrict__ out, uint32_t* __restrict__ tables,
                                                         const int32_t* __restrict__ perm, int slide, SchedArgs sched) {
  using PL = PairLds<C>;
  __shared__ __attribute__((aligned(16))) uint32_t lds[PL::WORDS];
  const Lane ln = make_lane<C>();
  uint32_t* gl = lds + ln.g * PL::STRIDE;
  constexpr int K2 = 2 * C::K;
  const int slot = blockIdx.x * C::GROUPS + ln.g;
  const int nslots = gridDim.x * C::GROUPS;
  const bool dual = base2.p != nullptr;
  const int TE = 1 << wb;
  uint32_t* tab = tables + (size_t)slot * (TE + (dual ? 16 : 0)) * K2;
  uint32_t* tab2 = tab + (size_t)TE * K2;
  const int nwin = (exp_words * 32 + wb - 1) / wb;
  const int nwin2 = dual ? exp2_words * 8 : 0;
  const int top_bit = (nwin - 1) * wb;
  const int words1 = base.words ? base.words : 2 * C::K32, words2 = base2.words ? base2.words : 2 * C::K32;
  WaveSched ws;                                                // mpe_sched.h: static trips, then (primaries only) the queue of tail units
  ws.init(sched);

#pragma unroll 1
  for (;;) {
    int ubase;
    if (!ws.next(sched, batch, nslots, C::GROUPS, ubase)) break;           // nothing left for this wave (wave-uniform): it leaves its SIMD to the others
#ifdef MPE_WAVE_TRACE
    const int trip = ws.trip - 1;
    const unsigned long long wt_m0 = __builtin_amdgcn_s_memtime(), wt_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int inst = ubase + ln.g;
    const bool active = inst < batch;
    const int pos = active ? inst : batch - 1;
    const int idx = (SLIDE && perm) ? perm[pos] : pos;        // perm: the launch's items ordered by key (sliding windows)
    const int mi = sel_index(mod_sel, idx);
    const uint32_t* ex = row_of(exps, idx);
    const uint32_t* ex2 = dual ? row_of(exps2, idx) : ex;
    // sliding windows: only when all the wave's exponentiations read the SAME exponent row (waves that straddle a key
    // boundary of the ordered launch keep the fixed windows) and, on a two-base ladder, the first window stays above exps2
    bool sl_rt = false;                                       // does THIS wave run the sliding schedule?
    int sw_lo = -1;                                           // the pending window multiplication: after the squaring that
    uint32_t sw_val = 0;                                      // brings b down to sw_lo, times tab[sw_val]
    UniformWords exu = nullptr;                               // the wave's exponent row through a scalar pointer: the window
    if (SLIDE && slide) {                                     // schedule stays in SGPRs and the phase machine wave-uniform
      const uint64_t ea = (uint64_t)(uintptr_t)ex;
      const uint32_t e_lo = __builtin_amdgcn_readfirstlane((uint32_t)ea), e_hi = __builtin_amdgcn_readfirstlane((uint32_t)(ea >> 32));
      sl_rt = __ballot(ea != (((uint64_t)e_hi << 32) | e_lo)) == 0;
      exu = (UniformWords)(((uint64_t)e_hi << 32) | e_lo);
      if (sl_rt) {
        sw_lo = slide_window(exu, exp_words, exp_words * 32 - 1, wb, sw_val);
        if (sw_lo < 0 || (dual && sw_lo < 32 * exp2_words)) sl_rt = false;
      }
    }

    uint32_t n[C::L];
    constexpr int KS = pair_kstore(C::BITS);
    load_owner<C>(n, ps.n_limbs + (size_t)mi * KS, ln);
    const uint32_t n0inv = ps.n0inv[mi];
    {
      const uint32_t* kc = ps.kc + (size_t)mi * KS;           // K_c + (R - 1), limb-wise, stays in LDS for the item
#pragma unroll
      for (int i = 0; i < C::L; ++i)                          // R - 1 is all-ones over STEPS limbs
        gl[PL::KC + ln.t + C::TPI * i] = kc[ln.t + C::TPI * i] + ((ln.t + C::TPI * i) < C::STEPS ? C::MASK : 0u);
      uint32_t t0[C::L], t1[C::L];                            // tab[0] = the form of 1
      load_owner<C>(t0, ps.one + (size_t)mi * 2 * KS, ln);
      load_owner<C>(t1, ps.one + (size_t)mi * 2 * KS + KS, ln);
      store_owner<C>(tab, t0, ln);
      store_owner<C>(tab + C::K, t1, ln);
    }

    const bool sl = SLIDE && sl_rt;                           // (the fixed-window instantiation carries none of this)
    uint32_t cur0[C::L], cur1[C::L];
    int which = 0;                                            // 0: base / tab, 1: base2 / tab2
    const uint32_t* bw = row_of(base, idx);
    int kin = (words1 + C::K32 - 1) / C::K32 - 1;             // chunk being absorbed (Horner from the top)
    load_words_as_limbs<C>(cur0, gl, bw + kin * C::K32, words1 - kin * C::K32, ln);
#pragma unroll
    for (int i = 0; i < C::L; ++i) cur1[i] = 0;
    int ph = kin > 0 ? PP_IN : PP_MONT;
    --kin;
    int k = 0, b = top_bit;
#pragma unroll 1
    while (ph != PP_DONE) {
      bool sq = false;
      // ---- multiplier pair -> LDS ----
      if (ph == PP_IN) {
        copy_pair_to_lds<C>(gl, ps.tp + (size_t)mi * 2 * KS, ln, KS);
      } else if (ph == PP_MONT) {
        copy_pair_to_lds<C>(gl, ps.r2 + (size_t)mi * 2 * KS, ln, KS);
      } else if (ph == PP_TAB) {
        // fixed windows: the multiplier is x (written once); sliding: it is x^2, left in place by PP_TSQ
        if (k == 1 && !(sl && !which)) { put_limbs<C>(gl + PL::B0, cur0, ln); put_limbs<C>(gl + PL::B1, cur1, ln); }
      } else if (ph == PP_SQ || (SLIDE && ph == PP_TSQ)) {
        put_limbs<C>(gl + PL::B0, cur0, ln);
#pragma unroll
        for (int i = 0; i < C::L; ++i) gl[PL::B1 + ln.t * C::L + i] = cur0[i] << 1;
        sq = true;
      } else if (ph == PP_MUL1) {
        const uint32_t w = sl ? sw_val : exp_window(ex, exp_words, b / wb, wb);
#ifndef MPE_NARROW_COPY                                     // (A/B switch: the word-by-word copy of rounds 1-3)
        copy_row_to_lds<C>(gl, tab + (size_t)w * K2, ln);
#else
        copy_pair_to_lds<C>(gl, tab + (size_t)w * K2, ln);
#endif
      } else if (ph == PP_MUL2) {
        const int wi = b >> 2;
        const uint32_t w = (ex2[wi >> 3] >> ((wi & 7) * 4)) & 15u;
#ifndef MPE_NARROW_COPY
        copy_row_to_lds<C>(gl, w ? tab2 + (size_t)w * K2 : tab, ln);
#else
        copy_pair_to_lds<C>(gl, w ? tab2 + (size_t)w * K2 : tab, ln);
#endif
      } else {                                                // PP_FINAL: times the plain pair (1, 0)
#pragma unroll
        for (int i = 0; i < C::L; ++i) {
          gl[PL::B0 + ln.t * C::L + i] = (ln.t == 0 && i == 0) ? 1u : 0u;
          gl[PL::B1 + ln.t * C::L + i] = 0u;
        }
      }
      wave_lds_sync();
      uint32_t r0[C::L], r1[C::L];
      pairmul<C>(r0, r1, cur0, cur1, gl, n, n0inv, sq, half != 0, ln);
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < C::L; ++i) { cur0[i] = r0[i]; cur1[i] = r1[i]; }

      // ---- product -> its place; next phase ----
      bool tables_done = false;
      if (ph == PP_IN) {
        // x0 += chunk kin, exactly normalised (the value stays a small multiple of N)
        uint32_t ch[C::L];
        load_words_as_limbs<C>(ch, gl, bw + kin * C::K32, C::K32, ln);
        int64_t z[C::L];
#pragma unroll
        for (int i = 0; i < C::L; ++i) z[i] = (int64_t)cur0[i] + (int64_t)ch[i];
        full_normalize<C>(z, ln);
#pragma unroll
        for (int i = 0; i < C::L; ++i) cur0[i] = (uint32_t)z[i];
        if (--kin < 0) ph = PP_MONT;
      } else if (ph == PP_MONT) {
        uint32_t* T = which ? tab2 : tab;
        store_owner<C>(T + K2, cur0, ln);
        store_owner<C>(T + K2 + C::K, cur1, ln);
        k = 1;
        ph = (sl && !which) ? PP_TSQ : PP_TAB;
      } else if (SLIDE && ph == PP_TSQ) {
        // sliding windows need the ODD powers only: x^2 becomes the multiplier of the table phase, x comes back from the table
        put_limbs<C>(gl + PL::B0, cur0, ln);
        put_limbs<C>(gl + PL::B1, cur1, ln);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        load_owner<C>(cur0, tab + K2, ln);
        load_owner<C>(cur1, tab + K2 + C::K, ln);
        ph = PP_TAB;
      } else if (ph == PP_TAB) {
        uint32_t* T = which ? tab2 : tab;
        const int step = (sl && !which) ? 2 : 1;              // x^3, x^5, ... x^(TE-1)  |  x^2, x^3, ...
        k += step;
        store_owner<C>(T + (size_t)k * K2, cur0, ln);
        store_owner<C>(T + (size_t)k * K2 + C::K, cur1, ln);
        if (k > (which ? 14 : TE - 2)) {
          if (!which && dual) {
            which = 1;
            bw = row_of(base2, idx);
            kin = (words2 + C::K32 - 1) / C::K32 - 1;
            load_words_as_limbs<C>(cur0, gl, bw + kin * C::K32, words2 - kin * C::K32, ln);
#pragma unroll
            for (int i = 0; i < C::L; ++i) cur1[i] = 0;
            ph = kin > 0 ? PP_IN : PP_MONT;
            --kin;
          } else {
            tables_done = true;
          }
        }
      } else if (ph == PP_FINAL) {
        ph = PP_DONE;
      } else {
        if (ph == PP_SQ) --b;
        if (ph == PP_MUL1 && sl) sw_lo = slide_window(exu, exp_words, b - 1, wb, sw_val);     // the next window below this one
        const bool m1 = ph == PP_SQ && (sl ? b == sw_lo : (b % wb) == 0);
        const bool m2 = ph != PP_MUL2 && dual && (b & 3) == 0 && (b >> 2) < nwin2;
        ph = m1 ? PP_MUL1 : (m2 ? PP_MUL2 : (b == 0 ? PP_FINAL : PP_SQ));
      }
      if (tables_done) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const uint32_t w = sl ? sw_val : exp_window(ex, exp_words, nwin - 1, wb);
        load_owner<C>(cur0, tab + (size_t)w * K2, ln);
        load_owner<C>(cur1, tab + (size_t)w * K2 + C::K, ln);
        b = sl ? sw_lo : top_bit;
        if (sl) sw_lo = slide_window(exu, exp_words, b - 1, wb, sw_val);
        ph = b == 0 ? PP_FINAL : PP_SQ;
      }
    }
    // canonical digits -> interface words z0 | z1.  The pair means the INTEGER z0 + z1 N with z0 < 2N lazily, so a
    // subtraction of N from z0 carries 1 into z1 (z0 >= N is a 2^-38 event for random operands, but e.g. the base N
    // itself lands exactly there); z1 is then reduced modulo N (multiples of N^2 drop out).
    {
      int64_t z[C::L];
#pragma unroll
      for (int i = 0; i < C::L; ++i) z[i] = (int64_t)cur0[i];
      full_normalize<C>(z, ln);
      const bool ge = cmp_ge<C>(z, n, ln);
      if (ge) {
#pragma unroll
        for (int i = 0; i < C::L; ++i) z[i] -= (int64_t)n[i];
        full_normalize<C>(z, ln);
      }
#pragma unroll
      for (int i = 0; i < C::L; ++i) { cur0[i] = (uint32_t)z[i]; z[i] = (int64_t)cur1[i]; }
      if (ge && !half && ln.t == 0) z[0] += 1;
      full_normalize<C>(z, ln);
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {                 // z1 + 1 <= 2N: at most two subtractions
        if (cmp_ge<C>(z, n, ln)) {
#pragma unroll
          for (int i = 0; i < C::L; ++i) z[i] -= (int64_t)n[i];
          full_normalize<C>(z, ln);
        }
      }
#pragma unroll
      for (int i = 0; i < C::L; ++i) cur1[i] = (uint32_t)z[i];
    }
    store_limbs_as_words<C>(out + (size_t)idx * 2 * C::K32, gl, cur0, active, ln);
    store_limbs_as_words<C>(out + (size_t)idx * 2 * C::K32 + C::K32, gl, cur1, active, ln);
#ifdef MPE_WAVE_TRACE
    if (g_wave_trace && threadIdx.x == 0) {
      const unsigned long long wt_m1 = __builtin_amdgcn_s_memtime(), wt_r1 = __builtin_amdgcn_s_memrealtime();
      const unsigned at = atomicAdd(&g_wave_trace_n, 1u);
      if (at < g_wave_trace_cap) {
        unsigned long long* w = g_wave_trace + (size_t)at * 8;
        w[0] = ((unsigned long long)gridDim.x << 32) | blockIdx.x;
        w[1] = ((unsigned long long)(unsigned)batch << 32) | ((unsigned)trip << 16) | ((unsigned)C::L << 8) | ((unsigned)(ws.role != 0) << 2) | ((unsigned)(half != 0) << 1) | (unsigned)(SLIDE ? 1 : 0);
        w[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID | HW_ID
        w[3] = wt_m0; w[4] = wt_m1; w[5] = wt_r0; w[6] = wt_r1;
        w[7] = ((unsigned long long)(unsigned)C::BITS << 32) | (unsigned)exp_words;
      }
    }
#endif
  }
}

// out[i] (2H words) = z0 + z1 * N   (z0 | z1 as left by pair_modexp_kernel; one item per lane)
template <int H>
__global__ void pair_finish_kernel(int B, Rows mod_sel, const uint32_t* __restrict__ mod_words, uint32_t* __restrict__ out) {
  MPE_FOREGROUND();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t z0[H], z1[H], nn[H], r[2 * H];
  sm::copy(z0, out + (size_t)i * 2 * H, H);
  sm::copy(z1, out + (size_t)i * 2 * H + H, H);
  sm::copy(nn, mod_words + (size_t)sel_index(mod_sel, i) * H, H);
  sm::mul(r, z1, H, nn, H);
  sm::add(r, 2 * H, r, 2 * H, z0, H);
  sm::copy(out + (size_t)i * 2 * H, r, 2 * H);
}

}  // namespace mpe

// host side: template implementations, instantiated by mpe_pair2048.hip / mpe_pair1024.hip ----------------
namespace mpe {

#ifdef MPE_WAVE_TRACE
// arm (buf = device memory for cap records of 8 x u64; nullptr switches the trace off) / read back the record count of THIS unit
static int wave_trace_arm_impl(void* buf, unsigned cap) {
  const unsigned zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave_trace), &buf, sizeof buf) != hipSuccess) return MPE_E_HIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave_trace_cap), &cap, sizeof cap) != hipSuccess) return MPE_E_HIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave_trace_n), &zero, sizeof zero) != hipSuccess) return MPE_E_HIP;
  return MPE_OK;
}
static int wave_trace_count_impl(unsigned* n) {
  return hipMemcpyFromSymbol(n, HIP_SYMBOL(g_wave_trace_n), sizeof *n) == hipSuccess ? MPE_OK : MPE_E_HIP;
}
#endif

template <class C>
int pairset_create_impl(int count, const uint32_t* d_moduli, mpe_pairset** out, hipStream_t st) {
  mpe_pairset* ps = new (std::nothrow) mpe_pairset();
  if (!ps) return MPE_E_NOMEM;
  ps->half_bits = C::BITS;
  ps->count = count;
  ps->mod_words = d_moduli;
  const size_t K = pair_kstore(C::BITS), words = (size_t)count * (K + 1 + 3 * 2 * K + K);
  static_assert(pair_kstore(C::BITS) >= C::K, "the stored constants must hold the layout's limbs");
  hipError_t e = hipMalloc(&ps->blob, words * 4);
  if (e != hipSuccess) { delete ps; mpe_set_error("hipMalloc(pairset)", e); return MPE_E_NOMEM; }
  (void)hipMemsetAsync(ps->blob, 0, words * 4, st);          // the padding limbs of every component are zero
  uint32_t* p = (uint32_t*)ps->blob;
  ps->n_limbs = p; p += count * K;
  ps->n0inv = p; p += count;
  ps->one = p; p += count * 2 * K;
  ps->r2 = p; p += count * 2 * K;
  ps->tp = p; p += count * 2 * K;
  ps->kc = p;
  hipLaunchKernelGGL(pairset_setup_kernel<C>, dim3((count + C::GROUPS - 1) / C::GROUPS), dim3(64), 0, st, count, d_moduli,
                     ps->n_limbs, ps->n0inv, ps->one, ps->r2, ps->tp, ps->kc);
  e = hipGetLastError();
  if (e != hipSuccess) { (void)hipFree(ps->blob); delete ps; mpe_set_error("pairset_setup_kernel", e); return MPE_E_HIP; }
  *out = ps;
  return MPE_OK;
}

template <class C>
int pair_modexp_impl(mpe_ctx* ctx, const mpe_pairset* ps, int batch, Rows mod_sel, Rows base, Rows exps, int exp_words,
                     Rows base2, Rows exps2, int exp2_words, int half, uint32_t* d_out, hipStream_t st, int public_exp) {
  const int units = (batch + C::GROUPS - 1) / C::GROUPS;
  const int grid = ladder_grid(ctx, units, ctx->cus * ctx->modexp_waves_per_cu);
  int wb = exp_words <= 8 ? 4 : (exp_words < 48 ? 5 : 6);
  if (ctx->window_bits) wb = ctx->window_bits;
  const bool dual = base2.p != nullptr;
  if (dual && (exp2_words <= 0 || 32 * exp2_words > ((exp_words * 32 + wb - 1) / wb - 1) * wb)) {
    mpe_set_error_msg("pair modexp: the second exponent must be shorter than the first");
    return MPE_E_ARG;
  }
  const size_t need = (size_t)grid * C::GROUPS * (((size_t)1 << wb) + (dual ? 16 : 0)) * 2 * C::K * sizeof(uint32_t);
  // A PUBLIC long exponent (the caller vouches: it is the key N) runs on sliding windows wherever a wave's 16 exponentiations
  // share it.  With several keys in the launch the items are first ordered by key (perm): three small kernels.
  const int slide = (public_exp && ctx->use_sliding && !half && exp_words >= 16 && wb >= 5) ? 1 : 0;
  const bool by_key = slide && ps->count > 1 && (mod_sel.idx || mod_sel.stride) && ps->count <= 4096 && batch > C::GROUPS;
  const size_t extra = (by_key ? ((size_t)batch + 2 * (size_t)ps->count + 64) * sizeof(int32_t) : 0) + SCHED_WORDS * sizeof(int32_t);
  uint32_t* tabs = tables_for(ctx, need + extra, st);
  if (!tabs) return MPE_E_NOMEM;
  SchedArgs sched = ladder_sched(ctx, units, ctx->cus * ctx->modexp_waves_per_cu, (int32_t*)((char*)tabs + need + extra) - SCHED_WORDS, st);
  if (sched.prio == 1) sched.prio = 2;          // the pair engine's launches are the stretches a small batch waits for (mpe_sched.h wave_priority)
  const int32_t* perm = nullptr;
  if (by_key) {
    int32_t* pm = (int32_t*)((char*)tabs + need);
    int32_t *cnt = pm + batch, *cursor = cnt + ps->count;
    (void)hipMemsetAsync(cnt, 0, (size_t)ps->count * sizeof(int32_t), st);
    const int nb = (batch + 255) / 256;
    hipLaunchKernelGGL(key_hist_kernel, dim3(nb), dim3(256), (size_t)ps->count * 4, st, batch, mod_sel, ps->count, cnt);
    hipLaunchKernelGGL(key_scan_kernel, dim3(1), dim3(64), 0, st, ps->count, cnt, cursor);
    hipLaunchKernelGGL(key_scatter_kernel, dim3(nb), dim3(256), (size_t)ps->count * 8, st, batch, mod_sel, ps->count, cursor, pm);
    perm = pm;
  }
  PairsetView v{ps->n_limbs, ps->n0inv, ps->one, ps->r2, ps->tp, ps->kc, ps->count};
  // kind 6: the pair kernel on sliding windows (the executed multiplication count differs: bench.py's accounting)
  prof_begin(ctx, st, half ? 4 : (slide ? 6 : 3), half ? C::BITS : 2 * C::BITS, exp_words, batch, dual ? exp2_words : 0);
  if (slide)
    hipLaunchKernelGGL((pair_modexp_kernel<C, true>), dim3(grid), dim3(64), 0, st, batch, v, mod_sel, base, exps, exp_words, wb, base2,
                       exps2, exp2_words, half, d_out, tabs, perm, slide, sched);
  else
    hipLaunchKernelGGL((pair_modexp_kernel<C, false>), dim3(grid), dim3(64), 0, st, batch, v, mod_sel, base, exps, exp_words, wb, base2,
                       exps2, exp2_words, half, d_out, tabs, perm, slide, sched);
  prof_end(ctx, st);
  if (slide) {
    // profiling only: "kind 6" says the launch was ALLOWED to slide; whether a wave does is decided at run time.  A counter
    // inside the hot kernel perturbs its register allocation (measured: +0.6 % kernel time, profiles/r04/ab_kernel_variants.json),
    // so the decision is REPLAYED by a one-thread-per-(wave, trip) audit kernel with the kernel's own rule and geometry.
    if (uint32_t* ctr = prof_counter(ctx, units))                                      // the units of the launch
      hipLaunchKernelGGL(pair_slide_audit_kernel, dim3(blocks_for(units, 64)), dim3(64), 0, st, batch, (int)C::GROUPS, exps,
                         exp_words, wb, dual ? exp2_words : 0, perm, ctr);
  }
  hipLaunchKernelGGL(pair_finish_kernel<C::K32>, dim3(blocks_for(batch, 64)), dim3(64), 0, st, batch, mod_sel, ps->mod_words, d_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("pair_modexp_kernel", e); return MPE_E_HIP; }
  ctx->last.waves = grid;
  ctx->last.ints_per_wave = C::GROUPS;
  ctx->last.limbs = 2 * C::K;
  ctx->last.limb_bits = C::W;
  ctx->last.lds_bytes_per_wave = PairLds<C>::WORDS * 4;
  ctx->last.table_scratch_bytes = need;
  return MPE_OK;
}

}  // namespace mpe
