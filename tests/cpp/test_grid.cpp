// Host-only check of the persistent-grid rule of the ladder kernels (multi_party_ecdsa_amd/csrc/mpe_internal.h: persistent_grid):
// whatever the mode, every wave slot of the grid gets at most `trips` groups and all `need` groups are covered; `equal` reproduces the
// rule of rounds 1-4; `full` launches all resident waves as soon as there is more than one pass; `hybrid` (the default) takes the
// full grid exactly when the tail is at most half a pass.  Built with `hipcc --cuda-host-only` by tests/test_grid_cpu.py (no GPU needed).
#include <cstdio>
#include <cstdlib>

#include "../../multi_party_ecdsa_amd/csrc/mpe_internal.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s (need %d cap %d mode %d grid %d)\n", #c, need, cap, mode, grid); return 1; } } while (0)

int main() {
  mpe_ctx ctx;
  long cases = 0;
  for (int cap : {2048, 1024, 8, 6}) {
    for (int mode = 0; mode < 3; ++mode) {
      ctx.grid_mode = mode;
      for (int need = 1; need <= 9 * cap + 3; ++need) {
        const int grid = mpe::persistent_grid(&ctx, need, cap);
        ++cases;
        CHECK(grid >= 1 && grid <= cap);
        const int trips = (need + grid - 1) / grid, min_trips = (need + cap - 1) / cap;
        CHECK(trips == min_trips);                                   // never more passes than the chip needs
        if (need <= cap) { CHECK(grid == need); continue; }
        const int rem = need % cap, equal = (need + min_trips - 1) / min_trips;
        if (rem == 0) { CHECK(grid == cap); continue; }
        if (mode == 0) CHECK(grid == equal && (long)(min_trips - 1) * grid < need);
        if (mode == 1) CHECK(grid == cap);
        if (mode == 2) CHECK(grid == (2 * rem <= cap ? cap : equal));
      }
    }
  }
  mpe_ctx dflt;
  if (dflt.grid_mode != 2) { std::printf("FAILED: the default is not hybrid\n"); return 1; }
  // round 6: the ladder kernels' scheduler (mpe_sched.h, ladder_grid / ladder_sched) — at most one unit per SIMD: election on twice the
  // workgroups; more units than resident waves: the whole chip and the unit queue; in between and under no_elect: static units.
  // (state = a dummy non-null pointer: ladder_sched only zeroes it through the HIP runtime when it uses it — host-only builds have no
  //  device, so the memset call fails harmlessly and its result is ignored)
  static int32_t dummy[mpe::SCHED_WORDS];
  for (int cap : {2048, 8}) {
    for (int need = 1; need <= 5 * cap + 1; ++need) {
      int mode = 0;
      mpe_ctx c;
      const int grid = mpe::ladder_grid(&c, need, cap);
      const mpe::SchedArgs a = mpe::ladder_sched(&c, need, cap, dummy, nullptr);
      ++cases;
      CHECK(a.units == need);
      if (2 * need <= cap) { CHECK(grid == 2 * need && a.state == dummy && a.mode == mpe::SCHED_PRIMARIES); }
      else if (need <= cap) { CHECK(grid == need && a.state == nullptr && a.mode == mpe::SCHED_STATIC); }
      else { CHECK(grid == cap && a.state == dummy && a.mode == mpe::SCHED_ALL); }
      mpe_ctx np; np.no_primaries = 1;
      const mpe::SchedArgs b = mpe::ladder_sched(&np, need, cap, dummy, nullptr);
      if (need <= cap) { CHECK(mpe::ladder_grid(&np, need, cap) == need && b.state == nullptr); } else { CHECK(b.mode == mpe::SCHED_ALL); }
      // wave priority of the launch (mpe_sched.h wave_priority): ladders 1, work started ahead of its round 0, nothing without use_prio
      CHECK(a.prio == 1);
      mpe_ctx bg; bg.ladder_prio = 0;
      CHECK(mpe::ladder_sched(&bg, need, cap, dummy, nullptr).prio == 0);
      mpe_ctx off; off.use_prio = 0; off.ladder_prio = 2;
      CHECK(mpe::ladder_sched(&off, need, cap, dummy, nullptr).prio == 0);
      mpe_ctx ne; ne.no_elect = 1;
      CHECK(mpe::ladder_grid(&ne, need, cap) == mpe::persistent_grid(&ne, need, cap) && mpe::ladder_sched(&ne, need, cap, dummy, nullptr).state == nullptr);
    }
  }
  std::printf("OK %ld cases\n", cases);
  return 0;
}
