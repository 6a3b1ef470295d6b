// RECORD of a round-4 experiment, removed from the shipped header in round 6 (not compiled by anything).
// Hand-issued LDS reads of the multiplier limbs in the CIOS passes of mpe_pairexp.h (`-DMPE_BQ`): correct, checked on the emitted
// ISA by check_bq_isa.py (beside this file), and NOT faster — 820.4 / 823.5 ms against 819.4 / 822.1 ms for the shipped kernel on
// the same box (profiles/r04/ab_kernel_variants.json, profiles/r04/README.md): the v_mad_u64_u32 stream already runs at the
// instruction's own issue rate, there is no idle slot to fill.  To revive it: paste the block below back in front of `pairmul`,
// call cios1q in place of cios1 where bq_layout_ok<C>(), and run check_bq_isa.py on the -save-temps ISA of both pair units.

// ---- the same passes with the multiplier limbs QUEUED two steps ahead (MPE_BQ) ---------------------------------------------------
// Where the waiting is (profiles/r03/pmc_stall_wave_cycles.json: 13.6 % of the wave cycles in s_waitcnt; the ISA of the loops
// above): hipcc issues the ds_read of a trip's first multiplier limbs at the TOP of the 18-step trip and waits for them three
// instructions later — a full LDS latency, 7 times per pass (more in the two-stream pass) — and when both waves of a SIMD sit
// there the multiplier idles (VALU port 87.5 % busy).  Source-level software pipelining does not survive the compiler (it sinks
// the loop-carried reads back to the loop top: tried, profiles/r04/README.md).  So the reads are issued BY HAND: limbs travel in
// pairs (one ds_read_b64 per two steps) through three 64-bit registers; at the first step of pair h the read of pair h+2 is
// issued and `s_waitcnt lgkmcnt(2)` lets exactly the two youngest reads stay in flight — LDS operations of a wave complete in
// order, so pair h has landed whatever else the compiler queued in between (its own ds_writes of the quotient digits only make
// the wait longer, never shorter).  The wait is tied to the destination register ("+v"), so no use can be scheduled above it.
// hipcc does not know the register is pending between the two statements: tools/check_bq_isa.py verifies in the emitted ISA that
// nothing reads, writes, copies or spills a destination between its ds_read and its wait (build.sh runs it).
template <int OFF>
__device__ __forceinline__ void bq_issue(uint64_t& d, uint32_t lds_addr) {
  // "+v": the destination is TIED to the register the previous pair of this slot lived in — one register per slot for the whole
  // pass, so the loop-carried value needs no copy (a copy of a register whose read is still in flight would copy stale bits)
  asm volatile("ds_read_b64 %0, %1 offset:%2 ; BQ_ISSUE %0" : "+v"(d) : "v"(lds_addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void bq_wait(uint64_t& d) {         // at most the N youngest LDS operations may still be in flight
  asm volatile("s_waitcnt lgkmcnt(%1) ; BQ_WAIT %0" : "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void bq_wait(uint64_t& d, uint64_t& e) {
  asm volatile("s_waitcnt lgkmcnt(%2) ; BQ_WAIT %0 ; BQ_WAIT %1" : "+v"(d), "+v"(e) : "n"(N));
}
__device__ __forceinline__ uint32_t lds_byte_address(const uint32_t* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)p;
}
template <class C>
constexpr bool bq_layout_ok() { return C::L % 6 == 0; }       // pairs of limbs, three registers: L/2 pairs per trip, a multiple of 3

template <class C, bool STORE_M>
__device__ __forceinline__ void cios1q(uint32_t (&res)[C::L], uint64_t (&c)[C::L], const uint32_t (&a)[C::L],
                                       const uint32_t* __restrict__ bl, uint32_t* __restrict__ ml,
                                       const uint32_t (&n)[C::L], uint32_t n0inv, const Lane& ln) {
  constexpr int L = C::L, W = C::W;
  static_assert(bq_layout_ok<C>(), "cios1q needs L % 6 == 0");
  uint32_t maskv = C::MASK;
  asm volatile("" : "+v"(maskv));
  uint64_t q0, q1, q2;                                         // pair h of a trip lives in q(h % 3)
  asm volatile("" : "=v"(q0), "=v"(q1), "=v"(q2));             // (defined, contents irrelevant: the first issue overwrites them)
  const uint32_t base = lds_byte_address(bl);
  bq_issue<0>(q0, base);
  bq_issue<8>(q1, base);
#pragma unroll 1
  for (int jj = 0;; ++jj) {
    uint32_t* mp = ml + jj * L;
    const uint32_t at = base + (uint32_t)(jj * L * 4);
    auto step = [&](auto rc) {
      constexpr int r = decltype(rc)::value;
      constexpr int h = r / 2;
      if constexpr (r % 2 == 0) {
        // pair h + 2 (pairs L/2 and L/2 + 1 are the first two of the next trip: the limbs are contiguous; in the last trip they are
        // reads past the multiplier, into the group's own LDS region — harmless, drained after the loop).  ALWAYS issued: every
        // slot is re-armed right after its last use, on every path, so its register is one unbroken live range
        if constexpr ((h + 2) % 3 == 0) bq_issue<8 * (h + 2)>(q0, at); else if constexpr ((h + 2) % 3 == 1) bq_issue<8 * (h + 2)>(q1, at); else bq_issue<8 * (h + 2)>(q2, at);
        // pair h is due: exactly two younger reads are in flight
        if constexpr (h % 3 == 0) bq_wait<2>(q0); else if constexpr (h % 3 == 1) bq_wait<2>(q1); else bq_wait<2>(q2);
      }
      const uint64_t qv = (h % 3 == 0) ? q0 : ((h % 3 == 1) ? q1 : q2);
      const uint32_t bj = (r & 1) ? (uint32_t)(qv >> 32) : (uint32_t)qv;
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
  // the reads the last pairs of the last trip asked for are still in flight: let them land before their registers are reused
  static_assert((C::STEPS % C::L) >= C::L - 2 || (C::STEPS % C::L) == 0, "the exit must lie in the trip's last pair");
  bq_wait<0>(q0, q1);
  bq_wait<0>(q2);
  cios_finish<C>(res, c, ln);
}

// (The two-stream pass has no queued form: six live 64-bit slots do not survive the register allocator at 256 VGPRs — the checker
// found copies of registers whose read was still in flight in every attempt, profiles/r04/README.md — so multiplications keep
// cios2 under MPE_BQ.)

