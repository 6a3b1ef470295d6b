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
  std::printf("OK %ld cases\n", cases);
  return 0;
}
