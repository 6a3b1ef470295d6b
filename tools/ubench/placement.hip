// Where does the dispatcher put the single-wave workgroups of a ladder-shaped launch (256 VGPRs: two waves per SIMD at most, 18.5 KB LDS)?
// Every workgroup records HW_ID / XCC_ID and then spins for a while so that the whole grid is co-resident; workgroups with
// blockIdx >= keep leave at once (the "tail" of a persistent grid: mpe_internal.h persistent_grid).  The host counts, among the kept
// workgroups, how many SIMDs hold one of them and how many hold two — the difference between a tail trip at ~0.5 and at 1.0 of a
// full trip (profiles/r05/ab_grid_three_modes.jsonl, ab_lanes.jsonl).
//
// build: hipcc --offload-arch=gfx950 -O2 -o placement placement.hip      run: ./placement > gpurun_out/placement.jsonl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
probe(int keep, unsigned spin_ticks, unsigned* __restrict__ hwid, unsigned* __restrict__ xcc, unsigned long long* __restrict__ t_start) {
  __shared__ unsigned lds[4640];                                       // 18 560 bytes: the pair kernel's LDS per wave
  asm volatile("v_mov_b32 v250, 0" ::: "v250");                        // claim the register budget of the ladder kernels
  const unsigned h = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
  const unsigned x = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  lds[threadIdx.x] = h;
  if (threadIdx.x == 0) { hwid[blockIdx.x] = h; xcc[blockIdx.x] = x & 15u; t_start[blockIdx.x] = t0; }
  if ((int)blockIdx.x >= keep) return;
  while ((unsigned)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (lds[(threadIdx.x + 1) & 63] == 0xFFFFFFFFu) hwid[blockIdx.x] = 0;
}

int main() {
  const int MAXG = 4096;
  unsigned *d_h, *d_x; unsigned long long* d_t;
  CK(hipMalloc(&d_h, MAXG * 4)); CK(hipMalloc(&d_x, MAXG * 4)); CK(hipMalloc(&d_t, MAXG * 8));
  std::vector<unsigned> h(MAXG), x(MAXG);
  struct Case { int grid, keep; };
  const Case cases[] = {{512, 512}, {768, 768}, {1024, 1024}, {1280, 1280}, {1536, 1536}, {2048, 2048}, {2048, 1024}, {2048, 768}, {2048, 512},
                        {2048, 1536}, {1024, 512}, {4096, 4096}};
  for (const Case& c : cases) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(probe, dim3(c.grid), dim3(64), 0, 0, c.keep, 200000u /* 2 ms */, d_h, d_x, d_t);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), d_h, c.grid * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(x.data(), d_x, c.grid * 4, hipMemcpyDeviceToHost));
      std::map<unsigned, int> per_simd, per_cu, per_xcc;
      const int kept = c.keep < c.grid ? c.keep : c.grid;
      for (int i = 0; i < kept; ++i) {
        const unsigned simd = (h[i] >> 4) & 3, cu = (h[i] >> 8) & 15, sh = (h[i] >> 12) & 1, se = (h[i] >> 13) & 7;
        const unsigned cu_key = (x[i] << 12) | (se << 8) | (sh << 4) | cu;
        ++per_simd[(cu_key << 2) | simd];
        ++per_cu[cu_key];
        ++per_xcc[x[i]];
      }
      int hist[9] = {0};
      for (auto& kv : per_simd) ++hist[kv.second < 8 ? kv.second : 8];
      int cu_hist[17] = {0};
      for (auto& kv : per_cu) ++cu_hist[kv.second < 16 ? kv.second : 16];
      printf("{\"grid\": %d, \"keep\": %d, \"rep\": %d, \"simds_used\": %zu, \"simds_with_1\": %d, \"simds_with_2\": %d, \"simds_with_3plus\": %d, "
             "\"cus_used\": %zu, \"xccs_used\": %zu, \"waves_per_cu_hist\": [",
             c.grid, c.keep, rep, per_simd.size(), hist[1], hist[2], hist[3] + hist[4] + hist[5] + hist[6] + hist[7] + hist[8], per_cu.size(),
             per_xcc.size());
      for (int k = 1; k <= 8; ++k) printf("%d%s", cu_hist[k], k < 8 ? ", " : "");
      // first workgroups: blockIdx -> (xcc, se, cu, simd)
      printf("], \"first16\": [");
      for (int i = 0; i < 16 && i < c.grid; ++i)
        printf("[%u, %u, %u, %u]%s", x[i], (h[i] >> 13) & 7, (h[i] >> 8) & 15, (h[i] >> 4) & 3, i < 15 && i + 1 < c.grid ? ", " : "");
      printf("]}\n");
    }
  }
  return 0;
}
