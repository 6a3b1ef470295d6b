// How long is one link of a dependent VALU chain for a LONE wave on gfx950 — the question behind the quotient-digit chain of the 9- and 5-limb
// ladder layouts (tools/model/lone_ladder_model.py: both sit at ~160 cycles per CIOS step whatever they issue).
// One wave (one workgroup of 64 lanes) runs `iters` trips of 16 copies of a chain body written with explicit registers; s_memtime brackets the
// loop.  Reported: shader-clock cycles per body and per link.  Bodies: single instructions feeding themselves, and the three candidate chains
// of one CIOS step for 16 lanes per integer — the shipped one, a 29-bit shadow accumulator (quotient lookahead), Orup's delayed-free quotient.
// build: hipcc --offload-arch=gfx950 -O3 -o chain_latency chain_latency.hip ; run: ./chain_latency > chain_latency.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","s40","s41","s42","s43","s44","s45","s46","s47","vcc","scc","memory"

#define X16(B) B B B B B B B B B B B B B B B B

#define KERNEL(NAME, BODY)                                                                    \
  __global__ void __launch_bounds__(64) NAME(unsigned long long* out, int iters) {            \
    unsigned long long t0, t1;                                                                \
    asm volatile(                                                                             \
      "s_mov_b32 s40, %2\n"                                                                   \
      "v_mov_b32 v32, 0x12345\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0x6789b\n v_mov_b32 v35, 0x1f2e3d\n v_mov_b32 v36, 0x7777\n v_mov_b32 v37, 0\n" \
      "v_mov_b32 v38, 0x1fffffff\n v_mov_b32 v39, 0\n v_mov_b32 v40, 3\n v_mov_b32 v41, 0\n v_mov_b32 v42, 5\n v_mov_b32 v43, 0\n"              \
      "v_mbcnt_lo_u32_b32 v44, -1, 0\n v_mbcnt_hi_u32_b32 v44, -1, v44\n v_lshlrev_b32 v44, 2, v44\n"                                            \
      "s_memtime %0\n s_waitcnt lgkmcnt(0)\n"                                                 \
      "1:\n" X16(BODY)                                                                        \
      "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"                     \
      "s_memtime %1\n s_waitcnt lgkmcnt(0)\n"                                                 \
      : "=&s"(t0), "=&s"(t1) : "s"(iters) : CLOB);                                            \
    if (threadIdx.x == 0) out[0] = t1 - t0;                                                   \
  }

// ---- single instructions feeding themselves -----------------------------------------------------------------------------------------------
KERNEL(k_mad_acc,   "v_mad_u64_u32 v[32:33], vcc, v34, v35, v[32:33]\n")                       // through the accumulator (src2)
KERNEL(k_mad_mul,   "v_mad_u64_u32 v[32:33], vcc, v32, v35, v[36:37]\n")                       // through the multiplier (src0)
KERNEL(k_mul_lo,    "v_mul_lo_u32 v32, v32, v35\n")
KERNEL(k_mul_hi,    "v_mul_hi_u32 v32, v32, v35\n")
KERNEL(k_and,       "v_and_b32 v32, v32, v38\n")
KERNEL(k_add,       "v_add_u32 v32, v32, v35\n")
KERNEL(k_lshl_add,  "v_lshl_add_u32 v32, v32, 3, v35\n")
KERNEL(k_add3,      "v_add3_u32 v32, v32, v35, v36\n")
KERNEL(k_lshr64,    "v_lshrrev_b64 v[32:33], 1, v[32:33]\n")
KERNEL(k_add64,     "v_lshl_add_u64 v[32:33], v[32:33], 0, v[36:37]\n")
KERNEL(k_alignbit,  "v_alignbit_b32 v32, v33, v32, 29\n")
KERNEL(k_dpp_quad,  "s_nop 1\n v_mov_b32_dpp v32, v32 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_shr4,  "s_nop 1\n v_mov_b32_dpp v32, v32 row_shr:4 row_mask:0xf bank_mask:0x2\n")
KERNEL(k_dpp_shr1,  "s_nop 1\n v_mov_b32_dpp v32, v32 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_nop1,      "s_nop 1\n")
KERNEL(k_swizzle,   "ds_swizzle_b32 v32, v32 offset:0x0010\n s_waitcnt lgkmcnt(0)\n")          // bit mode, and_mask 0x10: lane 0 of every 16
KERNEL(k_bpermute,  "ds_bpermute_b32 v32, v41, v32\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_readlane,  "v_readlane_b32 s44, v32, 0\n s_nop 3\n v_mov_b32 v32, s44\n")
KERNEL(k_lds_rt,    "ds_write_b32 v44, v32\n ds_read_b32 v32, v44\n s_waitcnt lgkmcnt(0)\n")   // a round trip through LDS memory

// ---- one CIOS step's chain, 16 lanes per integer ------------------------------------------------------------------------------------------
// shipped (mpe_pairexp.h cios1): c0 += a0 b_j -> lo(c0) n0inv -> three DPP hops -> mask -> c0 += m n0 -> c0 >> 29 -> c1 += ... (next step's c0)
KERNEL(k_step_shipped16,
       "v_mad_u64_u32 v[32:33], vcc, v34, v35, v[32:33]\n"
       "v_mul_lo_u32 v42, v36, v32\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 row_shr:4 row_mask:0xf bank_mask:0x2\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 row_shr:8 row_mask:0xf bank_mask:0xc\n"
       "v_and_b32 v42, v42, v38\n"
       "v_mad_u64_u32 v[32:33], vcc, v42, v40, v[32:33]\n"
       "v_lshrrev_b64 v[32:33], 29, v[32:33]\n"
       "v_lshl_add_u64 v[32:33], v[36:37], 0, v[32:33]\n")
// the same for 8 lanes per integer (two hops, the mask folded into the first)
KERNEL(k_step_shipped8,
       "v_mad_u64_u32 v[32:33], vcc, v34, v35, v[32:33]\n"
       "v_mul_lo_u32 v42, v36, v32\n"
       "s_nop 1\n v_and_b32_dpp v42, v42, v38 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 row_shr:4 row_mask:0xf bank_mask:0x2\n"
       "v_mad_u64_u32 v[32:33], vcc, v42, v40, v[32:33]\n"
       "v_lshrrev_b64 v[32:33], 29, v[32:33]\n"
       "v_lshl_add_u64 v[32:33], v[36:37], 0, v[32:33]\n")
// the same for 4 lanes per integer (one hop)
KERNEL(k_step_shipped4,
       "v_mad_u64_u32 v[32:33], vcc, v34, v35, v[32:33]\n"
       "v_mul_lo_u32 v42, v36, v32\n"
       "s_nop 1\n v_and_b32_dpp v42, v42, v38 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
       "v_mad_u64_u32 v[32:33], vcc, v42, v40, v[32:33]\n"
       "v_lshrrev_b64 v[32:33], 29, v[32:33]\n"
       "v_lshl_add_u64 v[32:33], v[36:37], 0, v[32:33]\n")
// quotient lookahead: T = 8 lo29(c0) kept as a 32-bit shadow; q8 = T n0inv -> hops -> H = hi(q8 n0) -> T' = (H << 3) + X (X off the chain)
KERNEL(k_step_look16,
       "v_mul_lo_u32 v42, v36, v32\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 row_shr:4 row_mask:0xf bank_mask:0x2\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 row_shr:8 row_mask:0xf bank_mask:0xc\n"
       "v_mul_hi_u32 v32, v42, v40\n"
       "v_lshl_add_u32 v32, v32, 3, v35\n")
KERNEL(k_step_look8,
       "v_mul_lo_u32 v42, v36, v32\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 row_shr:4 row_mask:0xf bank_mask:0x2\n"
       "v_mul_hi_u32 v32, v42, v40\n"
       "v_lshl_add_u32 v32, v32, 3, v35\n")
// Orup: q = lo29(c0) itself -> hops -> c1 += q Np0 -> (next step's c0)
KERNEL(k_step_orup16,
       "s_nop 1\n v_and_b32_dpp v42, v32, v38 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 row_shr:4 row_mask:0xf bank_mask:0x2\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 row_shr:8 row_mask:0xf bank_mask:0xc\n"
       "v_mad_u64_u32 v[32:33], vcc, v42, v40, v[36:37]\n")
KERNEL(k_step_orup8,
       "s_nop 1\n v_and_b32_dpp v42, v32, v38 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
       "s_nop 1\n v_mov_b32_dpp v42, v42 row_shr:4 row_mask:0xf bank_mask:0x2\n"
       "v_mad_u64_u32 v[32:33], vcc, v42, v40, v[36:37]\n")
// the broadcast through the LDS crossbar instead of three hops
KERNEL(k_step_look16_swz,
       "v_mul_lo_u32 v42, v36, v32\n"
       "ds_swizzle_b32 v42, v42 offset:0x0010\n s_waitcnt lgkmcnt(0)\n"
       "v_mul_hi_u32 v32, v42, v40\n"
       "v_lshl_add_u32 v32, v32, 3, v35\n")

struct E { const char* name; void (*fn)(unsigned long long*, int); int links; };
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  unsigned long long* out; CK(hipMalloc(&out, 64));
  const int iters = 4096;
  E es[] = {{"mad_u64_u32 via accumulator", k_mad_acc, 1}, {"mad_u64_u32 via multiplier", k_mad_mul, 1}, {"mul_lo_u32", k_mul_lo, 1}, {"mul_hi_u32", k_mul_hi, 1},
            {"and_b32", k_and, 1}, {"add_u32", k_add, 1}, {"lshl_add_u32", k_lshl_add, 1}, {"add3_u32", k_add3, 1}, {"lshrrev_b64", k_lshr64, 1},
            {"lshl_add_u64", k_add64, 1}, {"alignbit_b32", k_alignbit, 1}, {"s_nop 1 alone", k_nop1, 1}, {"s_nop 1 + mov_dpp quad_perm", k_dpp_quad, 1},
            {"s_nop 1 + mov_dpp row_shr:4", k_dpp_shr4, 1}, {"s_nop 1 + mov_dpp row_shr:1", k_dpp_shr1, 1}, {"ds_swizzle + wait", k_swizzle, 1},
            {"ds_bpermute + wait", k_bpermute, 1}, {"readlane + s_nop 3 + mov", k_readlane, 1}, {"ds_write + ds_read + wait", k_lds_rt, 1},
            {"step chain shipped, 16 lanes", k_step_shipped16, 9}, {"step chain shipped, 8 lanes", k_step_shipped8, 7}, {"step chain shipped, 4 lanes", k_step_shipped4, 6},
            {"step chain lookahead, 16 lanes", k_step_look16, 6}, {"step chain lookahead, 8 lanes", k_step_look8, 5},
            {"step chain Orup, 16 lanes", k_step_orup16, 4}, {"step chain Orup, 8 lanes", k_step_orup8, 3},
            {"step chain lookahead, 16 lanes, ds_swizzle", k_step_look16_swz, 4}};
  printf("{\"device\": \"%s\", \"bodies_per_run\": %d, \"results\": [\n", prop.gcnArchName, iters * 16);
  bool first = true;
  for (auto& e : es) {
    hipLaunchKernelGGL(e.fn, dim3(1), dim3(64), 0, 0, out, 64); CK(hipDeviceSynchronize());
    unsigned long long best = ~0ull;
    double best_ms = 1e30;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.fn, dim3(1), dim3(64), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c; CK(hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost));
      if (c < best) best = c;
      if (ms < best_ms) best_ms = ms;
    }
    const double per_body = (double)best / (iters * 16.0);
    printf("%s {\"chain\": \"%s\", \"links\": %d, \"memtime_ticks_per_body\": %.2f, \"per_link\": %.2f, \"wall_ns_per_body\": %.2f}", first ? "" : ",\n", e.name, e.links,
           per_body, per_body / e.links, best_ms * 1e6 / (iters * 16.0));
    first = false;
  }
  printf("\n]}\n");
  return 0;
}
