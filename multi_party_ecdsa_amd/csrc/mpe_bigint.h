// Wave-distributed reduced-radix big-integer arithmetic for gfx950 (MI355X).
//
// This is the engine under every modular exponentiation on the GG20 hot path (SURVEY.md §8a:
// curv `BigInt::mod_pow` -> GMP mpz_powm in the reference; call sites
// src/utilities/mta/range_proofs.rs:52-57,86,122-141 and
// src/utilities/zk_pdl_with_slack/mod.rs:189-195).
//
// Design (DESIGN.md "modexp kernel"):
//  * gfx950 issues v_mad_u64_u32 (32x32+64 -> 64) at the SAME rate as v_add_co_u32 / v_addc
//    (measured, profiles/r01_valu_rate.json: 16 lanes/clk/SIMD).  Carry instructions therefore
//    cost as much as multiplies, so operands are kept in a reduced radix 2^W (W=29) with one
//    64-bit column accumulator per limb: the inner loop is MACs only, no carry instructions.
//  * One big integer = K = TPI*L limbs spread over a group of TPI adjacent lanes (L limbs in
//    VGPRs per lane).  4096-bit: TPI=8, 2048-bit: TPI=4, both L=18 -> 8 / 16 integers per wave.
//  * CIOS Montgomery: per outer step every lane does L MACs with the broadcast multiplier limb
//    b_j (read from LDS), lane 0 of the group derives the quotient digit m, it is broadcast with
//    DPP, every lane does L MACs with m*n, and the accumulator shifts one limb: one 64-bit
//    column moves to the lower neighbour lane through DPP row_shl:1.
//  * Results stay "lazily normalised" (limbs < 2^W + 2^12, value < 2N) between multiplications.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpe {

// ---------------------------------------------------------------------------------------------
// configuration
// ---------------------------------------------------------------------------------------------
template <int BITS_, int W_, int L_, int TPI_>
struct Cfg {
  static constexpr int BITS = BITS_;        // width of the 32-bit-word interface (2048 / 4096)
  static constexpr int K32 = BITS_ / 32;    // interface words per integer
  static constexpr int W = W_;              // bits per internal limb
  static constexpr int L = L_;              // limbs per lane
  static constexpr int TPI = TPI_;          // lanes per integer
  static constexpr int K = L_ * TPI_;       // internal limbs per integer
  static constexpr int GROUPS = 64 / TPI_;  // integers per wave
  static constexpr uint32_t MASK = (1u << W_) - 1u;
  // outer CIOS steps of the N-adic pair engine (mpe_pairexp.h): R = 2^(W STEPS) only has to exceed 8N, and a multiplier
  // < 2N has no limbs beyond that — 71 of the 72 limbs for 2048-bit moduli (29 x 71 = 2059 bits), all 36 for 1024 bit
#ifdef MPE_FULL_STEPS                           // A/B switch: one step per limb, R = 2^(W K)
  static constexpr int STEPS = L_ * TPI_;
#else
  static constexpr int STEPS = (BITS_ + 3 + W_ - 1) / W_;
#endif
  static_assert(W_ * L_ * TPI_ >= BITS_ + 2, "R must exceed 4N");
  static_assert(K >= K32 + 2, "staging region reuse");
  // LDS words per group: K limbs + padding, chosen so the groups of one 32-lane half hit
  // distinct banks when they all read "their" b_j (ds_read_b32 banks = word index mod 32).
  static constexpr int pick_stride() {
    constexpr int per_half = (GROUPS >= 2) ? GROUPS / 2 : 1;
    for (int s = K + 2;; ++s) {
      bool ok = true;
      for (int a = 0; a < per_half && ok; ++a)
        for (int b = a + 1; b < per_half; ++b)
          if (((a * s) & 31) == ((b * s) & 31)) { ok = false; break; }
      if (ok) return s;
    }
  }
  static constexpr int STRIDE = pick_stride();
  static constexpr int LDS_WORDS = STRIDE * GROUPS;
};

// Radix choice.  Every column is split each time it reaches a lane's lowest position (every L
// steps), so a column only has to absorb 2L products (+ a small fold carry) between splits:
//   2L * 2^(2W + 0.01) < 2^64   ->   W = 29 with L = 18 (tools/model/montmul_model.py checks the
// worst case: max column 2^63).  W = 29 needs K = 144 limbs for 4096 bit (TPI = 8) and K = 72
// for 2048 bit (TPI = 4).  W = 27 / L = 19 is kept as a build-time alternative for A/B runs.
#ifndef MPE_W
#define MPE_W 29
#endif
#ifndef MPE_L
#define MPE_L 18
#endif
using Cfg4096 = Cfg<4096, MPE_W, MPE_L, 8>;
using Cfg2048 = Cfg<2048, MPE_W, MPE_L, 4>;
using Cfg1024 = Cfg<1024, MPE_W, MPE_L, 2>;     // the halves of the N-adic arithmetic modulo p^2 | q^2 (mpe_pairexp.h)

// ---------------------------------------------------------------------------------------------
// cross-lane primitives (DPP; VALU only, no LDS traffic)
// ---------------------------------------------------------------------------------------------
// lane i <- lane i+1 within a row of 16 (row_shl:1); lane 15 of a row reads 0 (bound_ctrl).
__device__ __forceinline__ uint32_t pull_next(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true);
}
// lane i <- lane i-1 within a row of 16 (row_shr:1); lane 0 of a row reads 0.
__device__ __forceinline__ uint32_t pull_prev(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
}
__device__ __forceinline__ uint64_t pull_next64(uint64_t x) {
  return ((uint64_t)pull_next((uint32_t)(x >> 32)) << 32) | pull_next((uint32_t)x);
}
__device__ __forceinline__ uint64_t pull_prev64(uint64_t x) {
  return ((uint64_t)pull_prev((uint32_t)(x >> 32)) << 32) | pull_prev((uint32_t)x);
}
// broadcast the value held by lane 0 of each TPI-lane group to the whole group.
template <int TPI>
__device__ __forceinline__ uint32_t bcast0(uint32_t x) {
  if constexpr (TPI == 1) {
    return x;
  } else if constexpr (TPI == 2) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xA0, 0xf, 0xf, true);  // quad_perm [0,0,2,2]
  } else {
    int y = __builtin_amdgcn_update_dpp(0, (int)x, 0x00, 0xf, 0xf, true);           // quad_perm [0,0,0,0]
    if constexpr (TPI == 8) {
      // lanes 4-7 / 12-15 of each row take lanes 0-3 / 8-11 (row_shr:4, banks 1 and 3 only)
      y = __builtin_amdgcn_update_dpp(y, y, 0x114, 0xf, 0xA, false);
    } else if constexpr (TPI == 16) {
      y = __builtin_amdgcn_update_dpp(y, y, 0x114, 0xf, 0x2, false);                // 4-7 <- 0-3
      y = __builtin_amdgcn_update_dpp(y, y, 0x118, 0xf, 0xC, false);                // 8-15 <- 0-7
    }
    return (uint32_t)y;
  }
}
// bcast0(x) & MASK with the mask applied between the two DPP stages, so that the first stage and the mask fold
// into one v_and_b32_dpp (the VALU issue slot is what the modexp kernel is bound by)
// (maskv holds the mask in a VGPR: DPP encodings take no literal operand)
template <int TPI>
__device__ __forceinline__ uint32_t bcast0_masked(uint32_t x, uint32_t maskv) {
  if constexpr (TPI == 8) {
    int y = __builtin_amdgcn_update_dpp(0, (int)x, 0x00, 0xf, 0xf, true) & (int)maskv;  // quad_perm [0,0,0,0]
    y = __builtin_amdgcn_update_dpp(y, y, 0x114, 0xf, 0xA, false);                      // 4-7 <- 0-3 of each half row
    return (uint32_t)y;
  } else if constexpr (TPI == 4) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x00, 0xf, 0xf, true) & maskv;
  } else if constexpr (TPI == 2) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xA0, 0xf, 0xf, true) & maskv;   // quad_perm [0,0,2,2]
  } else {
    return bcast0<TPI>(x) & maskv;
  }
}
// Order LDS traffic between the lanes of one wave (lanes exchange operands through LDS; the
// hardware executes one wave's DS ops in order, this only stops the compiler reordering them).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// per-lane view of the group
// ---------------------------------------------------------------------------------------------
struct Lane {
  int lane;    // 0..63
  int t;       // lane within group
  int g;       // group within wave
  bool t0;     // t == 0
};
template <class C>
__device__ __forceinline__ Lane make_lane() {
  Lane ln;
  ln.lane = threadIdx.x & 63;
  ln.t = ln.lane % C::TPI;
  ln.g = ln.lane / C::TPI;
  ln.t0 = (ln.t == 0);
  return ln;
}

// ---------------------------------------------------------------------------------------------
// Montgomery multiplication  res = a * b * 2^(-W*K) mod n   (value < 2n, lazily normalised)
//   a   : this lane's L limbs (limbs < 2^W + 2^12)
//   bl  : the group's K multiplier limbs in LDS (limbs < 2^W + 2^12)
//   n   : this lane's L limbs of the modulus (exactly normalised), n0inv = -n^-1 mod 2^W
// ---------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void montmul(uint32_t (&res)[C::L], const uint32_t (&a)[C::L],
                                        const uint32_t* __restrict__ bl, const uint32_t (&n)[C::L],
                                        uint32_t n0inv, const Lane& ln) {
  constexpr int L = C::L, W = C::W;
  uint64_t c[L];
#pragma unroll
  for (int i = 0; i < L; ++i) c[i] = 0;
  uint32_t maskv = C::MASK;
  asm volatile("" : "+v"(maskv));                  // keep the mask in a register (see bcast0_masked)

#pragma unroll 1
  for (int jj = 0; jj < C::TPI; ++jj) {
    const uint32_t* bp = bl + jj * L;
    // L outer steps, fully unrolled: step r keeps logical column i in physical c[(r+i)%L],
    // so the one-limb shift per step is a renaming and only the incoming top column moves.
#pragma unroll
    for (int r = 0; r < L; ++r) {
      const uint32_t bj = bp[r];
      c[r] += (uint64_t)a[0] * bj;
      const uint32_t m = bcast0_masked<C::TPI>((uint32_t)c[r] * n0inv, maskv);
#pragma unroll
      for (int i = 1; i < L; ++i) c[(r + i) % L] += (uint64_t)a[i] * bj;
#pragma unroll
      for (int i = 0; i < L; ++i) c[(r + i) % L] += (uint64_t)m * n[i];
      // One-limb shift.  Every lane splits its lowest column: the part above 2^W stays with the
      // lane (it has the weight of the next column), the low W bits move to the lower neighbour
      // as its new top column.  For lane 0 of the group the low part is 0 by construction of m,
      // so the previous group's top lane (and lane 15 of a row, via bound_ctrl) pulls in a zero.
      c[(r + 1) % L] += c[r] >> W;
      c[r] = (uint64_t)(pull_next((uint32_t)c[r]) & maskv);      // mask after the move: folds into one v_and_b32_dpp
    }
  }
  // local ripple, then hand the lane's carry-out (< 2^38) to the next lane without rippling on
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

// ---------------------------------------------------------------------------------------------
// exact helpers (used once per exponentiation / per modulus set-up; not hot)
// ---------------------------------------------------------------------------------------------
// Full carry propagation across the group.  Limbs are signed so this also settles borrows.
template <class C>
__device__ __forceinline__ void full_normalize(int64_t (&x)[C::L], const Lane& ln) {
  int64_t cin = 0;
#pragma unroll 1
  for (int pass = 0; pass <= C::TPI; ++pass) {
    int64_t carry = cin;
#pragma unroll
    for (int i = 0; i < C::L; ++i) {
      const int64_t v = x[i] + carry;
      x[i] = v & (int64_t)C::MASK;
      carry = v >> C::W;                 // arithmetic shift: borrows travel as negative carries
    }
    cin = (int64_t)pull_prev64((uint64_t)carry);
    if (ln.t0) cin = 0;
  }
}
// x >= n for exactly normalised x, n (both distributed).  Lane t holds more significant limbs
// than lane t-1, so comparing the per-lane verdict bit-masks as integers orders the groups.
template <class C>
__device__ __forceinline__ bool cmp_ge(const int64_t (&x)[C::L], const uint32_t (&n)[C::L], const Lane& ln) {
  bool gt = false, lt = false;
#pragma unroll
  for (int i = C::L - 1; i >= 0; --i) {
    const uint32_t xi = (uint32_t)x[i];
    if (!gt && !lt) {
      if (xi > n[i]) gt = true;
      else if (xi < n[i]) lt = true;
    }
  }
  const uint64_t G = __ballot(gt), Lm = __ballot(lt);
  const int sh = ln.g * C::TPI;
  const uint64_t gm = (C::TPI == 64) ? ~0ull : ((1ull << C::TPI) - 1ull);
  return ((G >> sh) & gm) >= ((Lm >> sh) & gm);
}
// lazily normalised value < 2n  ->  exactly normalised value in [0, n)
template <class C>
__device__ __forceinline__ void reduce_once(uint32_t (&v)[C::L], const uint32_t (&n)[C::L], const Lane& ln) {
  int64_t x[C::L];
#pragma unroll
  for (int i = 0; i < C::L; ++i) x[i] = (int64_t)v[i];
  full_normalize<C>(x, ln);
  if (cmp_ge<C>(x, n, ln)) {
#pragma unroll
    for (int i = 0; i < C::L; ++i) x[i] -= (int64_t)n[i];
    full_normalize<C>(x, ln);
  }
#pragma unroll
  for (int i = 0; i < C::L; ++i) v[i] = (uint32_t)x[i];
}

// Interface words (little-endian u32, K32 of them, zero-padded up to bit W*K + 32, in LDS) -> limbs.
template <class C>
__device__ __forceinline__ void limbs_from_words(uint32_t (&v)[C::L], const uint32_t* w32, const Lane& ln) {
  // The per-limb word indices and shifts depend on the lane only; laundering t keeps the compiler from hoisting
  // those ~36 values out of the exponentiation loop and holding them in VGPRs for the whole kernel.
  int t = ln.t;
  asm volatile("" : "+v"(t));
#pragma unroll
  for (int i = 0; i < C::L; ++i) {
    const int bitpos = (t * C::L + i) * C::W;
    const int q = bitpos >> 5, s = bitpos & 31;
    const uint64_t two = (uint64_t)w32[q] | ((uint64_t)w32[q + 1] << 32);
    v[i] = (uint32_t)(two >> s) & C::MASK;
  }
}
// Exactly normalised limbs (K of them followed by >= 2 zero words, in LDS) -> interface word q.
template <class C>
__device__ __forceinline__ uint32_t word_from_limbs(const uint32_t* xl, int q) {
  const int bitpos = q * 32;
  const int p = bitpos / C::W, s = bitpos - p * C::W;
  const uint64_t lo2 = (uint64_t)xl[p] | ((uint64_t)xl[p + 1] << C::W);
  uint32_t w = (uint32_t)(lo2 >> s);
  if (2 * C::W - s < 32) w |= xl[p + 2] << (2 * C::W - s);
  return w;
}

// Stage one K32-word integer from global memory into the group's LDS region (coalesced per
// group) and zero the padding words; the whole wave must call wave_lds_sync() afterwards.
template <class C>
__device__ __forceinline__ void stage_words(uint32_t* gl, const uint32_t* __restrict__ src, const Lane& ln,
                                            int nwords = C::K32) {
  // limbs_from_words reads words q, q+1 for bit positions up to W*K - 1 (> BITS): zero the tail
  constexpr int LAST = (C::W * C::K - 1) / 32 + 1;
  static_assert(LAST < C::STRIDE, "staging padding must fit the group's LDS region");
  for (int q = ln.t; q <= LAST; q += C::TPI) gl[q] = q < nwords ? src[q] : 0u;
}
// Write this lane's limbs into the group's LDS region in limb order (the "b" operand layout).
template <class C>
__device__ __forceinline__ void put_limbs(uint32_t* gl, const uint32_t (&v)[C::L], const Lane& ln) {
#pragma unroll
  for (int i = 0; i < C::L; ++i) gl[ln.t * C::L + i] = v[i];
}

}  // namespace mpe
