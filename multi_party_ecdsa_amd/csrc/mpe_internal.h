// Internal definitions shared by the translation units of libmpecdsa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <new>
#include <string>

#include "../../include/mpecdsa_hip.h"
#include "mpe_bigint.h"

struct mpe_ctx {
  int device = 0;
  int cus = 256;
  int modexp_waves_per_cu = 8;    // 2 waves/SIMD: the montmul loop holds ~230 VGPRs and already issues back-to-back
  void* tables = nullptr;         // window-table scratch, grown on demand
  size_t tables_bytes = 0;
  mpe_launch_info last = {};
};

struct mpe_modset {
  int bits = 0;
  int count = 0;
  int K = 0;
  void* blob = nullptr;
  uint32_t* n_limbs = nullptr;    // [count][K]   modulus, internal radix
  uint32_t* one_limbs = nullptr;  // [count][K]   R mod n
  uint32_t* r2_limbs = nullptr;   // [count][K]   R^2 mod n
  uint32_t* n0inv = nullptr;      // [count]      -n^-1 mod 2^W
};

void mpe_set_error(const char* what, hipError_t e);
