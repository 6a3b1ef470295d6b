// Does s_setprio let a YOUNGER wave win the issue arbitration of a SIMD against an older one?  (mpe_sched.h: without it the arbiter serves the
// older wave first — a ladder wave keeps 0.92 of its lone speed, whatever joins it later runs in what is left.)
// Two launches of 1 024 single-wave workgroups with the register budget of the ladder kernels (two waves per SIMD at most), on two streams:
// launch A first, launch B 200 us later, so that every SIMD holds one A wave (older) and one B wave (younger).  Each wave sets its priority
// (0..3), runs the same fixed stream of dependent-free v_mad_u64_u32, and records HW_ID and its own duration (s_memrealtime).  Reported per
// (prio A, prio B): the median duration of the A and of the B waves that really shared a SIMD with a wave of the other launch, against the
// duration of a wave alone.
// build: hipcc --offload-arch=gfx950 -O3 -o setprio setprio.hip ; run: ./setprio > setprio.json
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

#define M8 "v_mad_u64_u32 v[32:33], vcc, v48, v49, v[32:33]\n v_mad_u64_u32 v[34:35], vcc, v48, v49, v[34:35]\n" \
           "v_mad_u64_u32 v[36:37], vcc, v48, v49, v[36:37]\n v_mad_u64_u32 v[38:39], vcc, v48, v49, v[38:39]\n" \
           "v_mad_u64_u32 v[40:41], vcc, v48, v49, v[40:41]\n v_mad_u64_u32 v[42:43], vcc, v48, v49, v[42:43]\n" \
           "v_mad_u64_u32 v[44:45], vcc, v48, v49, v[44:45]\n v_mad_u64_u32 v[46:47], vcc, v48, v49, v[46:47]\n"

template <int PRIO>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
work(int iters, unsigned* __restrict__ hwid, unsigned* __restrict__ xcc, unsigned long long* __restrict__ t0s, unsigned long long* __restrict__ t1s) {
  asm volatile("v_mov_b32 v250, 0" ::: "v250");                        // claim the register budget of the ladder kernels
  __builtin_amdgcn_s_setprio(PRIO);
  const unsigned h = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
  const unsigned x = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  asm volatile(
      "s_mov_b32 s40, %0\n v_mov_b32 v48, 0x12345\n v_mov_b32 v49, 0x6789b\n"
      "1:\n" M8 M8 M8 M8
      "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"
      : : "s"(iters) : "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","s40","vcc","scc");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { hwid[blockIdx.x] = h; xcc[blockIdx.x] = x & 15u; t0s[blockIdx.x] = t0; t1s[blockIdx.x] = t1; }
}

__global__ void delay_kernel(unsigned ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((unsigned)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
}

typedef void (*Kern)(int, unsigned*, unsigned*, unsigned long long*, unsigned long long*);
static Kern kern_of(int p) { return p == 0 ? work<0> : (p == 1 ? work<1> : (p == 2 ? work<2> : work<3>)); }

static double median(std::vector<double>& v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const int G = 1024, iters = 40000;                                   // 32 x 40 000 multiply-adds per wave: ~3 ms alone
  unsigned *d_h[2], *d_x[2]; unsigned long long *d_t0[2], *d_t1[2];
  for (int k = 0; k < 2; ++k) { CK(hipMalloc(&d_h[k], G * 4)); CK(hipMalloc(&d_x[k], G * 4)); CK(hipMalloc(&d_t0[k], G * 8)); CK(hipMalloc(&d_t1[k], G * 8)); }
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  std::vector<unsigned> h[2], x[2]; std::vector<unsigned long long> t0[2], t1[2];
  for (int k = 0; k < 2; ++k) { h[k].resize(G); x[k].resize(G); t0[k].resize(G); t1[k].resize(G); }
  auto fetch = [&](int k) {
    CK(hipMemcpy(h[k].data(), d_h[k], G * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(x[k].data(), d_x[k], G * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(t0[k].data(), d_t0[k], G * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(t1[k].data(), d_t1[k], G * 8, hipMemcpyDeviceToHost));
  };
  auto simd_key = [&](int k, int i) { return (x[k][i] << 16) | (((h[k][i] >> 13) & 7) << 12) | (((h[k][i] >> 12) & 1) << 11) | (((h[k][i] >> 8) & 15) << 4) | ((h[k][i] >> 4) & 3); };
  // alone
  hipLaunchKernelGGL(kern_of(0), dim3(G), dim3(64), 0, sa, iters, d_h[0], d_x[0], d_t0[0], d_t1[0]); CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(kern_of(0), dim3(G), dim3(64), 0, sa, iters, d_h[0], d_x[0], d_t0[0], d_t1[0]); CK(hipDeviceSynchronize());
  fetch(0);
  std::vector<double> alone; std::map<unsigned, int> cnt;
  for (int i = 0; i < G; ++i) ++cnt[simd_key(0, i)];
  for (int i = 0; i < G; ++i) if (cnt[simd_key(0, i)] == 1) alone.push_back((t1[0][i] - t0[0][i]) * 10.0);
  const double alone_ns = median(alone);
  printf("{\"waves_per_launch\": %d, \"alone_us\": %.1f, \"alone_waves_on_their_own_simd\": %zu, \"pairs\": [\n", G, alone_ns / 1e3, alone.size());
  bool first = true;
  const int combos[][2] = {{0, 0}, {0, 3}, {3, 0}, {0, 1}, {1, 0}, {2, 3}, {3, 3}};
  for (auto& pc : combos) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kern_of(pc[0]), dim3(G), dim3(64), 0, sa, iters, d_h[0], d_x[0], d_t0[0], d_t1[0]);
      hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, sb, 20000u);                                       // 200 us: A's waves are resident first
      hipLaunchKernelGGL(kern_of(pc[1]), dim3(G), dim3(64), 0, sb, iters, d_h[1], d_x[1], d_t0[1], d_t1[1]);
      CK(hipDeviceSynchronize());
      fetch(0); fetch(1);
      std::map<unsigned, int> ca, cb;
      for (int i = 0; i < G; ++i) { ++ca[simd_key(0, i)]; ++cb[simd_key(1, i)]; }
      std::vector<double> da, db;
      int shared = 0;
      for (int i = 0; i < G; ++i) if (ca[simd_key(0, i)] == 1 && cb.count(simd_key(0, i)) && cb[simd_key(0, i)] == 1) { da.push_back((t1[0][i] - t0[0][i]) * 10.0); ++shared; }
      for (int i = 0; i < G; ++i) if (cb[simd_key(1, i)] == 1 && ca.count(simd_key(1, i)) && ca[simd_key(1, i)] == 1) db.push_back((t1[1][i] - t0[1][i]) * 10.0);
      const double ma = median(da), mb = median(db);
      printf("%s {\"prio_older\": %d, \"prio_younger\": %d, \"rep\": %d, \"simds_with_one_of_each\": %d, \"older_us\": %.1f, \"younger_us\": %.1f, "
             "\"older_speed_vs_alone\": %.3f, \"younger_speed_vs_alone\": %.3f}", first ? "" : ",\n", pc[0], pc[1], rep, shared, ma / 1e3, mb / 1e3,
             ma > 0 ? alone_ns / ma : 0.0, mb > 0 ? alone_ns / mb : 0.0);
      first = false;
    }
  }
  printf("\n]}\n");
  return 0;
}
