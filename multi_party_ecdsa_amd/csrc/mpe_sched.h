// Which wave runs which unit of a ladder launch (a "unit" = the C::GROUPS exponentiations one wave carries through one ladder).
//
// What the per-wave trace of round 6 showed (profiles/r06/wave_trace.jsonl, tools/trace_waves.py: HW_ID, s_memtime, s_memrealtime of
// every (wave, trip) of pair_modexp_kernel):
//   * a SIMD does not share fairly: the OLDER of two resident ladder waves runs at 0.92 of its lone speed and the younger one in
//     what is left (a 2048-bit ladder: 35 ms alone; 38 and 64 ms for two that start together; three units on one SIMD: 88 ms).
//     Two waves per SIMD deliver 1.19x the units of one — throughput launches keep two — but a launch lasts as long as its
//     SLOWEST wave, so a launch that has at most one unit per SIMD must put exactly one wave on each;
//   * the dispatcher does not guarantee that.  A launch of <= 1 024 single-wave workgroups lands one per SIMD when the chip is
//     idle and quiet (round 5's probe), but behind the tail of a previous kernel, or beside kernels of a forked stream, 5 - 25 % of
//     its waves share a SIMD (1 024 waves: 978 alone + 23 pairs; inside a signing step: 424 alone + 44 pairs of 512, 792 + 116 pairs
//     of 1 024) — and the pairs set the launch's time: 58.9 ms instead of 35.6, 40.4 instead of 28.  That — placement, not clocks
//     (2.35 - 2.40 GHz for lone waves, 2.10 - 2.25 GHz with every SIMD doubled) nor issue arbitration of a lone wave — is the
//     "fresh launch of lone waves at 0.95 of a shared trip" round 5 could not explain, and why the TAIL of a full grid looked
//     different: there every SIMD already held exactly two waves.
// Two consequences, both about WHO runs WHICH unit:
//   * a launch with at most one unit per SIMD (units <= half the resident waves) takes placement out of the dispatcher's hands.
//     It starts up to TWICE as many workgroups as it has units; every wave reads HW_ID / XCC_ID and draws a ticket of ITS SIMD (one
//     atomic): the first arrival on a SIMD is that SIMD's PRIMARY, everybody else a secondary.  Primaries take units from a queue
//     (one atomic per unit), so no SIMD runs two units of the launch side by side.  A secondary holds its slot while units are
//     still unclaimed — the dispatcher then has to put the workgroups it has not placed yet on OTHER SIMDs — and leaves when the
//     queue is empty; if units are STILL unclaimed when the whole grid has arrived (SIMDs held by kernels of a forked stream never
//     got a workgroup) it takes one itself: two units on that SIMD, 38 + 64 ms, instead of a primary running two in a row.  ("Arrived"
//     counts a primary only AFTER its first pull: counted on arrival, a late primary lost its unit to an early secondary of another SIMD
//     and the launch doubled up everywhere — 60 ms.  A
//     50 us time-out instead of "the whole grid has arrived" fired while the dispatcher was still placing workgroups: 1 - 9 SIMDs
//     of 1 024 doubled up and the launch took 59 ms instead of 35.7, profiles/r06/wave_trace_sched2.jsonl);
//   * a launch of several passes hands out EVERY unit from the queue, to every wave (MODE_ALL).  With static units (trip x grid +
//     block, rounds 1-5) the favoured wave of a SIMD finished its units at 0.63 of the launch and left, and the other one ran the rest
//     alone at the lone rate; pulling keeps both resident until the queue is dry — the favoured wave simply takes more units —
//     and a tail (n.f passes) balances itself the same way.
// MODE_STATIC (state == nullptr) behaves exactly as in rounds 1-5.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpe {

constexpr int SCHED_SIMD_IDS = 8192;                 // (xcc 0..7, se 0..7, sh 0..1, cu 0..15, simd 0..3) -> a dense id
constexpr int SCHED_WORDS = 2 + SCHED_SIMD_IDS;      // [0] waves arrived, [1] next tail unit, [2 + id] arrivals on that SIMD

enum { SCHED_STATIC = 0, SCHED_ALL = 1, SCHED_PRIMARIES = 2 };
struct SchedArgs {
  int32_t* state;       // SCHED_WORDS zeroed words, or nullptr: static units (trip * nslots + blockIdx.x * GROUPS)
  int mode;             // SCHED_ALL: every wave pulls units from the queue; SCHED_PRIMARIES: the first arrival of every SIMD does
  int units;            // units of the launch (queue length)
  int prio;             // s_setprio of the launch's waves (0 = the hardware's default): see wave_priority below
};

// Wave priority (s_setprio, 0..3).  Among waves of EQUAL priority a SIMD serves the older one first (the trace above); a wave of HIGHER
// priority wins whatever its age — measured (tools/ubench/setprio.hip, profiles/r06/setprio.json: two multiply-add streams on one SIMD):
//     equal priorities        older 0.95 - 1.0 of its lone speed, younger what is left (0.37 while both run)
//     younger at 1, older 0   younger 1.0, older what is left — any gap of priority does it (1 against 0 like 3 against 0)
// Used by the small-batch schedule (mpe_gg20.h): a launch that is NOT on the critical path (the PDL proofs' beta^N started two rounds
// ahead) runs at 0 beside short kernels that raise themselves to 1 and ladders of the path at 2.  The immediate of s_setprio is a literal:
#define MPE_FOREGROUND() __builtin_amdgcn_s_setprio(1)     /* first statement of the short kernels of rounds 2 and 3 */
__device__ __forceinline__ void wave_priority(int prio) {
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio == 3) __builtin_amdgcn_s_setprio(3);
}

__device__ __forceinline__ int sched_simd_id() {
  const unsigned h = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: simd 5:4, cu 11:8, sh 12, se 15:13
  const unsigned x = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;    // HW_REG_XCC_ID 3:0
  return (int)((((x * 8u + ((h >> 13) & 7u)) * 2u + ((h >> 12) & 1u)) * 16u + ((h >> 8) & 15u)) * 4u + ((h >> 4) & 3u));
}

struct WaveSched {
  int trip = 0;
  int role = 0;         // 0: this SIMD's primary (or a launch without election), > 0: a later arrival
  bool first_pull = false;
  __device__ __forceinline__ void init(const SchedArgs& a) {
    wave_priority(a.prio);
    if (!a.state || a.mode != SCHED_PRIMARIES) return;
    int r = 0;
    if (threadIdx.x == 0) {
      r = atomicAdd(a.state + 2 + sched_simd_id(), 1);
      if (r != 0) atomicAdd(a.state, 1);               // a secondary has "arrived" now; a primary only after its first pull (next())
    }
    role = __builtin_amdgcn_readfirstlane(r);
    first_pull = role == 0;
  }
  __device__ __forceinline__ int pull(const SchedArgs& a) {
    int u = 0;
    if (threadIdx.x == 0) u = atomicAdd(a.state + 1, 1);
    return __builtin_amdgcn_readfirstlane(u);
  }
  // first item position of this wave's next unit; false: nothing left for this wave (wave-uniform)
  __device__ __forceinline__ bool next(const SchedArgs& a, int batch, int nslots, int groups, int& ubase) {
    if (!a.state) {
      ubase = trip * nslots + (int)blockIdx.x * groups;
      ++trip;
      return ubase < batch;
    }
    if (a.mode == SCHED_PRIMARIES && role != 0) {
      // a secondary: stay (asleep) while units are unclaimed and workgroups of the grid are still to come — each of them may be the
      // primary of a SIMD that has none yet; leave when the queue is dry; help out when the whole grid has arrived and units are
      // STILL unclaimed (no primary will come for them), or after 10 ms (the rest of the grid is stuck behind other kernels)
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        if (__hip_atomic_load(a.state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.units) return false;
        if (__hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (int)gridDim.x) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 >= 1000000ull) break;
        __builtin_amdgcn_s_sleep(32);
      }
      role = 0;                                        // from now on it pulls like a primary
    }
    const int u = pull(a);
    if (first_pull) {                                  // (SCHED_PRIMARIES) state[0] == gridDim.x now means: every primary HAS taken its first unit
      first_pull = false;
      if (threadIdx.x == 0) atomicAdd(a.state, 1);
    }
    if (u >= a.units) return false;
    ubase = u * groups;
    ++trip;
    return true;
  }
};

}  // namespace mpe
