// gfx950 VALU/DS instruction issue-rate microbenchmark.
//
// the 32-bit integer multiply family's rate on gfx950 is not in the
// guides; the modexp kernel's roofline ("peak_imad_per_s") is whatever this probe measures.
// Every candidate instruction is issued from inline asm (8 independent chains x UNROLL),
// timed with hipEvents (wall) and s_memtime (shader cycles), at 1/2/4/8 waves per SIMD.
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
// run  : ./valu_rate > gpurun_out/valu_rate.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;   // loop trips
constexpr int UNROLL = 32;    // instructions per trip (4 rounds over 8 chains)

// X(name, asm-body using %0..%7 as 32-bit regs a0..a7, %8..%15 as 64-bit regs d0..d7, %16 = sgpr s, %17 = vgpr b)
#define BODY8(INS) \
  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

#define DEFKERNEL(NAME, ASMSTR)                                                              \
  __global__ void __launch_bounds__(256) k_##NAME(unsigned* out, unsigned long long* cyc,    \
                                                   unsigned seed) {                           \
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u,    \
             a3 = a1 * 31u, a4 = a2 ^ 0x1234567u, a5 = a3 + 99u, a6 = a4 * 17u, a7 = a5 ^ a6; \
    unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;\
    double f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;           \
    unsigned b = a0 | 1u;                                                                     \
    unsigned s = __builtin_amdgcn_readfirstlane(seed | 3u);                                   \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int it = 0; it < ITERS; ++it) {                                                      \
      asm volatile(ASMSTR                                                                     \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),   \
                     "+v"(a7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5),   \
                     "+v"(d6), "+v"(d7), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4),   \
                     "+v"(f5), "+v"(f6), "+v"(f7)                                             \
                   : "s"(s), "v"(b)                                                           \
                   : "vcc", "s40", "s41", "s42", "s43", "memory");                           \
    }                                                                                         \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
    unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^  \
                 d5 ^ d6 ^ d7) ^ (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);           \
    if (r == 0x12345u) out[0] = r;                                                            \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
  }

// operand indices: a_i = %i (0..7), d_i = %(8+i), f_i = %(16+i), s = %24, b = %25
#define STR(x) #x
#define A(i) "%" STR(i)
// helpers to compute 8+i / 16+i textually
#define D0 "%8"
#define D1 "%9"
#define D2 "%10"
#define D3 "%11"
#define D4 "%12"
#define D5 "%13"
#define D6 "%14"
#define D7 "%15"
#define F0 "%16"
#define F1 "%17"
#define F2 "%18"
#define F3 "%19"
#define F4 "%20"
#define F5 "%21"
#define F6 "%22"
#define F7 "%23"
#define S_ "%24"
#define B_ "%25"
#define DD(i) D##i
#define FF(i) F##i

#define I_MAD64(i) "v_mad_u64_u32 " DD(i) ", vcc, " A(i) ", " B_ ", " DD(i) "\n"
#define I_MAD64S(i) "v_mad_u64_u32 " DD(i) ", vcc, " A(i) ", " S_ ", " DD(i) "\n"
#define I_MAD64SS(i) "v_mad_u64_u32 " DD(i) ", s[40:41], " A(i) ", " S_ ", " DD(i) "\n"
#define I_MULLO(i) "v_mul_lo_u32 " A(i) ", " A(i) ", " B_ "\n"
#define I_MULHI(i) "v_mul_hi_u32 " A(i) ", " A(i) ", " B_ "\n"
#define I_MAD24(i) "v_mad_u32_u24 " A(i) ", " A(i) ", " B_ ", " A(i) "\n"
#define I_MUL24(i) "v_mul_u32_u24 " A(i) ", " A(i) ", " B_ "\n"
#define I_MULHI24(i) "v_mul_hi_u32_u24 " A(i) ", " A(i) ", " B_ "\n"
#define I_MADU16(i) "v_mad_u32_u16 " A(i) ", " A(i) ", " B_ ", " A(i) "\n"
#define I_DOT2(i) "v_dot2_u32_u16 " A(i) ", " A(i) ", " B_ ", " A(i) "\n"
#define I_DOT4(i) "v_dot4_u32_u8 " A(i) ", " A(i) ", " B_ ", " A(i) "\n"
#define I_ADD(i) "v_add_u32 " A(i) ", " A(i) ", " B_ "\n"
#define I_ADDCO(i) "v_add_co_u32 " A(i) ", vcc, " A(i) ", " B_ "\n"
#define I_ADDC(i) "v_addc_co_u32 " A(i) ", vcc, " A(i) ", " B_ ", vcc\n"
#define I_ADD3(i) "v_add3_u32 " A(i) ", " A(i) ", " B_ ", " S_ "\n"
#define I_LSHLADD(i) "v_lshl_add_u32 " A(i) ", " A(i) ", 3, " B_ "\n"
#define I_ALIGNBIT(i) "v_alignbit_b32 " A(i) ", " A(i) ", " B_ ", 28\n"
#define I_AND(i) "v_and_b32 " A(i) ", " A(i) ", " B_ "\n"
#define I_LSHLADD64(i) "v_lshl_add_u64 " DD(i) ", " DD(i) ", 0, " DD(i) "\n"
#define I_LSHR64(i) "v_lshrrev_b64 " DD(i) ", 28, " DD(i) "\n"
#define I_FMA32(i) "v_fma_f32 " A(i) ", " A(i) ", " B_ ", " A(i) "\n"
#define I_FMA64(i) "v_fma_f64 " FF(i) ", " FF(i) ", " FF(i) ", " FF(i) "\n"
#define I_PKFMA32(i) "v_pk_fma_f32 " DD(i) ", " DD(i) ", " DD(i) ", " DD(i) "\n"
#define I_MOVDPP_WSHR(i) "v_mov_b32_dpp " A(i) ", " A(i) " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MOVDPP_RSHR(i) "v_mov_b32_dpp " A(i) ", " A(i) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP_RSHR(i) "v_add_u32_dpp " A(i) ", " A(i) ", " A(i) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_READLANE(i) "v_readlane_b32 s42, " A(i) ", 5\n"
#define I_READLANE_USE(i) "v_readlane_b32 s42, " A(i) ", 5\nv_add_u32 " A(i) ", s42, " A(i) "\n"
#define I_SWIZZLE(i) "ds_swizzle_b32 " A(i) ", " A(i) " offset:swizzle(BITMASK_PERM, \"00000\")\n"
#define I_BPERM(i) "ds_bpermute_b32 " A(i) ", " B_ ", " A(i) "\n"
#define I_PERMLANE32(i) "v_permlane32_swap_b32 " A(i) ", " B_ "\n"

#define WAIT_LGKM "s_waitcnt lgkmcnt(0)\n"

DEFKERNEL(mad_u64_u32_vv, BODY8(I_MAD64))
DEFKERNEL(mad_u64_u32_vs, BODY8(I_MAD64S))
DEFKERNEL(mad_u64_u32_vs_scc, BODY8(I_MAD64SS))
DEFKERNEL(mul_lo_u32, BODY8(I_MULLO))
DEFKERNEL(mul_hi_u32, BODY8(I_MULHI))
DEFKERNEL(mad_u32_u24, BODY8(I_MAD24))
DEFKERNEL(mul_u32_u24, BODY8(I_MUL24))
DEFKERNEL(mul_hi_u32_u24, BODY8(I_MULHI24))
DEFKERNEL(mad_u32_u16, BODY8(I_MADU16))
DEFKERNEL(dot2_u32_u16, BODY8(I_DOT2))
DEFKERNEL(dot4_u32_u8, BODY8(I_DOT4))
DEFKERNEL(add_u32, BODY8(I_ADD))
DEFKERNEL(add_co_u32, BODY8(I_ADDCO))
DEFKERNEL(addc_co_u32, BODY8(I_ADDC))
DEFKERNEL(add3_u32, BODY8(I_ADD3))
DEFKERNEL(lshl_add_u32, BODY8(I_LSHLADD))
DEFKERNEL(alignbit_b32, BODY8(I_ALIGNBIT))
DEFKERNEL(and_b32, BODY8(I_AND))
DEFKERNEL(lshl_add_u64, BODY8(I_LSHLADD64))
DEFKERNEL(lshrrev_b64, BODY8(I_LSHR64))
DEFKERNEL(fma_f32, BODY8(I_FMA32))
DEFKERNEL(fma_f64, BODY8(I_FMA64))
DEFKERNEL(pk_fma_f32, BODY8(I_PKFMA32))
DEFKERNEL(mov_dpp_wave_shr1, BODY8(I_MOVDPP_WSHR))
DEFKERNEL(mov_dpp_row_shr1, BODY8(I_MOVDPP_RSHR))
DEFKERNEL(add_dpp_row_shr1, BODY8(I_ADDDPP_RSHR))
DEFKERNEL(readlane, BODY8(I_READLANE))
DEFKERNEL(readlane_use, BODY8(I_READLANE_USE))
DEFKERNEL(ds_swizzle, BODY8(I_SWIZZLE) WAIT_LGKM)
DEFKERNEL(ds_bpermute, BODY8(I_BPERM) WAIT_LGKM)
DEFKERNEL(permlane32_swap, BODY8(I_PERMLANE32))

// mixed: 4 mads + 4 adds interleaved (does a full-rate op hide under a multi-pass op of the same wave?)
#define I_MIX(i) "v_mad_u64_u32 " DD(i) ", vcc, " A(i) ", " S_ ", " DD(i) "\nv_add_u32 " A(i) ", " A(i) ", " B_ "\n"
DEFKERNEL(mix_mad64_add, BODY8(I_MIX))
// mad + ds_swizzle interleaved (DS pipe is separate from VALU)
#define I_MIXDS(i) "v_mad_u64_u32 " DD(i) ", vcc, " B_ ", " S_ ", " DD(i) "\nds_swizzle_b32 " A(i) ", " A(i) " offset:swizzle(BITMASK_PERM, \"00000\")\n"
DEFKERNEL(mix_mad64_swizzle, BODY8(I_MIXDS) WAIT_LGKM)

struct Entry { const char* name; void (*fn)(unsigned*, unsigned long long*, unsigned); int ninstr_per_slot; };
#define E(NAME, N) { #NAME, k_##NAME, N }
static Entry entries[] = {
  E(fma_f32, 1), E(pk_fma_f32, 1), E(fma_f64, 1),
  E(mad_u64_u32_vv, 1), E(mad_u64_u32_vs, 1), E(mad_u64_u32_vs_scc, 1), E(mul_lo_u32, 1), E(mul_hi_u32, 1),
  E(mad_u32_u24, 1), E(mul_u32_u24, 1), E(mul_hi_u32_u24, 1), E(mad_u32_u16, 1), E(dot2_u32_u16, 1),
  E(dot4_u32_u8, 1), E(add_u32, 1), E(add_co_u32, 1), E(addc_co_u32, 1), E(add3_u32, 1),
  E(lshl_add_u32, 1), E(alignbit_b32, 1), E(and_b32, 1), E(lshl_add_u64, 1), E(lshrrev_b64, 1),
  E(mov_dpp_wave_shr1, 1), E(mov_dpp_row_shr1, 1), E(add_dpp_row_shr1, 1), E(readlane, 1),
  E(readlane_use, 2), E(ds_swizzle, 1), E(ds_bpermute, 1), E(permlane32_swap, 1),
  E(mix_mad64_add, 2), E(mix_mad64_swizzle, 2),
};

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double clk_ghz = prop.clockRate / 1e6;
  unsigned* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, sizeof(unsigned long long) * cus * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"iters\": %d, \"unroll\": %d, \"results\": [\n",
         prop.gcnArchName, cus, clk_ghz, ITERS, UNROLL);
  bool first = true;
  for (auto& en : entries) {
    for (int wps : {1, 2, 4, 8}) {   // waves per SIMD (256-thread blocks = 1 wave per SIMD each)
      int grid = cus * wps;
      hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), 0, 0, out, cyc, 1u);  // warm
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), 0, 0, out, cyc, 1u + rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      std::vector<unsigned long long> h(grid);
      CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
      double winstr_per_wave = (double)ITERS * UNROLL * en.ninstr_per_slot;
      // shader cycles per wave-instruction per SIMD (wps waves share a SIMD)
      double cyc_per_instr_simd = avg / (winstr_per_wave * wps);
      double total_lane_ops = winstr_per_wave * 64.0 * 4.0 * grid;     // 4 waves per block
      double lane_ops_per_s = total_lane_ops / (best * 1e-3);
      printf("%s{\"instr\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"memtime_cycles_per_instr_per_simd\": %.3f, "
             "\"lane_ops_per_s\": %.4e, \"wall_cycles_per_instr_per_simd_at_2.4GHz\": %.3f}",
             first ? "" : ",\n", en.name, wps, best, cyc_per_instr_simd, lane_ops_per_s,
             (best * 1e-3 * 2.4e9) / (winstr_per_wave * wps));
      first = false;
    }
  }
  printf("\n]}\n");
  return 0;
}
