// Does the issue rate of v_mad_u64_u32 on gfx950 depend on WHERE its four source dwords live in the VGPR file?
// (src0, src1: one dword each; src2: an aligned pair; VGPR bank = index mod 4.)  Every variant below is the same
// instruction stream -- 16 independent accumulator chains, 16 multiplicands, one broadcast multiplier -- with different
// register assignments, written with explicit registers.  Timed by wall clock over ~2 M wave-instructions per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o mad_banks mad_banks.hip ; run: ./mad_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55", \
  "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
  "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103", \
  "v104","v105","v106","v107","v108","v109","v110","v111","s40","s41","s42","s43","vcc","scc","memory"

// M(acc_lo, a, b, cout): acc pair v[acc_lo:acc_lo+1] += v[a] * v[b]
#define M(ACC, ACC1, A_, B_, CO) "v_mad_u64_u32 v[" #ACC ":" #ACC1 "], " CO ", v" #A_ ", v" #B_ ", v[" #ACC ":" #ACC1 "]\n"

#define KERNEL(NAME, BODY)                                                                    \
  __global__ void __launch_bounds__(64) NAME(unsigned* out, int iters) {                      \
    asm volatile(                                                                             \
      "s_mov_b32 s40, %0\n"                                                                   \
      "v_mov_b32 v100, 0x12345\n v_mov_b32 v101, 0x6789b\n v_mov_b32 v102, 0x1f2e3d\n v_mov_b32 v103, 0x7777\n" \
      "1:\n" BODY BODY BODY BODY                                                              \
      "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"                     \
      : : "s"(iters) : CLOB);                                                                 \
    if (iters < 0) out[0] = 1;                                                                \
  }

// conflict-free: accumulators in banks {0,1}, multiplicands in bank 2, multiplier in bank 3
#define FREE(CO) \
  M(32,33,34,103,CO) M(36,37,38,103,CO) M(40,41,42,103,CO) M(44,45,46,103,CO) M(48,49,50,103,CO) M(52,53,54,103,CO) M(56,57,58,103,CO) M(60,61,62,103,CO) \
  M(64,65,66,103,CO) M(68,69,70,103,CO) M(72,73,74,103,CO) M(76,77,78,103,CO) M(80,81,82,103,CO) M(84,85,86,103,CO) M(88,89,90,103,CO) M(92,93,94,103,CO)
// multiplier in the bank of the accumulators' low word
#define BCONF(CO) \
  M(32,33,34,100,CO) M(36,37,38,100,CO) M(40,41,42,100,CO) M(44,45,46,100,CO) M(48,49,50,100,CO) M(52,53,54,100,CO) M(56,57,58,100,CO) M(60,61,62,100,CO) \
  M(64,65,66,100,CO) M(68,69,70,100,CO) M(72,73,74,100,CO) M(76,77,78,100,CO) M(80,81,82,100,CO) M(84,85,86,100,CO) M(88,89,90,100,CO) M(92,93,94,100,CO)
// multiplicand and multiplier in the same bank (2), accumulators in {0,1}
#define ABCONF(CO) \
  M(32,33,34,102,CO) M(36,37,38,102,CO) M(40,41,42,102,CO) M(44,45,46,102,CO) M(48,49,50,102,CO) M(52,53,54,102,CO) M(56,57,58,102,CO) M(60,61,62,102,CO) \
  M(64,65,66,102,CO) M(68,69,70,102,CO) M(72,73,74,102,CO) M(76,77,78,102,CO) M(80,81,82,102,CO) M(84,85,86,102,CO) M(88,89,90,102,CO) M(92,93,94,102,CO)
// what a compiler does: accumulators packed (pairs alternate {0,1} / {2,3}), multiplicands packed, multiplier anywhere
#define PACKED(CO) \
  M(32,33,64,103,CO) M(34,35,65,103,CO) M(36,37,66,103,CO) M(38,39,67,103,CO) M(40,41,68,103,CO) M(42,43,69,103,CO) M(44,45,70,103,CO) M(46,47,71,103,CO) \
  M(48,49,72,103,CO) M(50,51,73,103,CO) M(52,53,74,103,CO) M(54,55,75,103,CO) M(56,57,76,103,CO) M(58,59,77,103,CO) M(60,61,78,103,CO) M(62,63,79,103,CO)
// everything in banks {0,1}: multiplicand bank 0, multiplier bank 0
#define WORST(CO) \
  M(32,33,96,100,CO) M(36,37,96,100,CO) M(40,41,96,100,CO) M(44,45,96,100,CO) M(48,49,96,100,CO) M(52,53,96,100,CO) M(56,57,96,100,CO) M(60,61,96,100,CO) \
  M(64,65,96,100,CO) M(68,69,96,100,CO) M(72,73,96,100,CO) M(76,77,96,100,CO) M(80,81,96,100,CO) M(84,85,96,100,CO) M(88,89,96,100,CO) M(92,93,96,100,CO)

KERNEL(k_free_vcc, FREE("vcc"))
KERNEL(k_free_sgpr, FREE("s[42:43]"))
KERNEL(k_bconf_sgpr, BCONF("s[42:43]"))
KERNEL(k_abconf_sgpr, ABCONF("s[42:43]"))
KERNEL(k_packed_vcc, PACKED("vcc"))
KERNEL(k_packed_sgpr, PACKED("s[42:43]"))
KERNEL(k_worst_sgpr, WORST("s[42:43]"))

struct E { const char* name; void (*fn)(unsigned*, int); };
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, iters = 32768;          // 64 instructions per trip: 2.1 M per wave
  unsigned* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  E es[] = {{"free_vcc", k_free_vcc}, {"free_sgpr", k_free_sgpr}, {"bconf_sgpr", k_bconf_sgpr}, {"abconf_sgpr", k_abconf_sgpr},
            {"packed_vcc", k_packed_vcc}, {"packed_sgpr", k_packed_sgpr}, {"worst_sgpr", k_worst_sgpr}};
  printf("{\"device\": \"%s\", \"cus\": %d, \"instr_per_wave\": %d, \"results\": [\n", prop.gcnArchName, cus, iters * 64);
  bool first = true;
  for (auto& e : es) for (int wps : {1, 2, 4}) {
    const int grid = cus * 4 * wps;
    hipLaunchKernelGGL(e.fn, dim3(grid), dim3(64), 0, 0, out, 256); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(e.fn, dim3(grid), dim3(64), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double instr = (double)iters * 64, lane_ops = instr * 64 * grid / (best * 1e-3);
    printf("%s {\"variant\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"T_lane_ops_per_s\": %.2f, \"cycles_per_instr_per_simd_at_2.4GHz\": %.3f}",
           first ? "" : ",\n", e.name, wps, best, lane_ops / 1e12, best * 1e-3 * 2.4e9 / (instr * wps));
    first = false;
  }
  printf("\n]}\n");
  return 0;
}
