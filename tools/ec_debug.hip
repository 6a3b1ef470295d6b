// debug aid: exercise the pieces of r2b_kernel one by one (run with a short timeout)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../multi_party_ecdsa_amd/csrc/mpe_ec.h"
using namespace mpe;
__global__ void k_oncurve(uint32_t* out) {
  const ec::Aff H = ec::aff_h2();
  const ec::U256 l = ec::fe_sqr(H.y);
  ec::U256 seven = ec::u256_zero(); seven.w[0] = 7;
  const ec::U256 r = ec::fe_add(ec::fe_mul(ec::fe_sqr(H.x), H.x), seven);
  out[0] = ec::u256_eq(l, r) ? 1 : 0;
}
__global__ void k_mulh(uint32_t* out) {
  ec::U256 k = ec::u256_zero(); k.w[0] = 12345; k.w[3] = 0x9abcdef1; k.w[7] = 0x7fffffff;
  const ec::Aff r = ec::jac_to_aff(ec::jac_mul(k, ec::aff_h2()));
  for (int i = 0; i < 8; ++i) { out[i] = r.x.w[i]; out[8 + i] = r.y.w[i]; }
}
__global__ void k_hash(uint32_t* out) {
  const ec::Aff G = ec::aff_gen(), H = ec::aff_h2();
  const ec::Aff hp[5] = {G, H, G, H, G};
  ec::Sha256 s; ec::sha_init(s);
  for (int i = 0; i < 5; ++i) ec::sha_point_uncompressed(s, hp[i]);
  const ec::U256 d = ec::sha_final(s);
  const ec::U256 e = ec::sc_reduce(d.w, 8);
  for (int i = 0; i < 8; ++i) out[i] = e.w[i];
}
__global__ void k_scalar(uint32_t* out) {
  ec::U256 a = ec::u256_zero(), b = ec::u256_zero();
  for (int i = 0; i < 8; ++i) { a.w[i] = 0xfffffff0u + i; b.w[i] = 0x89abcdefu ^ (i * 77u); }
  a = ec::sc_reduce(a.w, 8); b = ec::sc_reduce(b.w, 8);
  const ec::U256 c = ec::sc_add(a, ec::sc_mul(a, b));
  const ec::U256 d = ec::sc_inv(c);
  const ec::U256 e = ec::sc_mul(c, d);
  for (int i = 0; i < 8; ++i) out[i] = e.w[i];
}
int main() {
  uint32_t* d; hipMalloc(&d, 256); uint32_t h[16];
  auto run = [&](const char* name, void (*k)(uint32_t*)) {
    hipMemset(d, 0, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("%s: %s :", name, hipGetErrorString(e));
    for (int i = 0; i < 16; ++i) printf(" %08x", h[i]);
    printf("\n"); fflush(stdout);
  };
  run("oncurve", k_oncurve); run("scalar(expect 1,0..)", k_scalar); run("hash", k_hash); run("mulh", k_mulh);
  return 0;
}
