// secp256k1 field / scalar / point arithmetic and SHA-256, one item per lane (plain u32[8] values).
//
// Replaces, for the GG20 hot path, what the reference reaches through curv-kzen's
// `Point<Secp256k1>` / `Scalar<Secp256k1>` (libsecp256k1 underneath) and `sha2::Sha256`:
// call sites src/utilities/mta/mod.rs:147-148,166-171, src/utilities/zk_pdl_with_slack/mod.rs:86,
// 102-110,138-142, src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:546-936.
// One item per lane.  The base field lives in mpe_fe.h (10 x 26-bit limbs, lazy reduction: no carry chains inside a
// multiplication), the Jacobian formulas and the window ladders in mpe_jac.h; this file keeps the 32-bit word helpers,
// the scalar field (mod q, 8 x 32-bit words: cold), the comb tables of the two fixed generators and SHA-256.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mpecdsa_hip.h"
#include "mpe_jac.h"
#include "mpe_sc.h"

namespace mpe {
namespace ec {

// ---- points: mpe_jac.h; here the fixed generators and their comb tables ----------------------------------
__device__ __forceinline__ Aff aff_gen() { Aff g; g.x = u256_load(GX); g.y = u256_load(GY); g.inf = false; return g; }
__device__ __forceinline__ Aff aff_h2() { Aff g; g.x = u256_load(H2X); g.y = u256_load(H2Y); g.inf = false; return g; }
__device__ __forceinline__ Aff aff_neg(const Aff& a) {
  Aff r = a;
  if (!a.inf && !u256_is_zero(a.y)) u256_sub(r.y, u256_load(FP), a.y);
  return r;
}
// Comb tables of the two fixed generators: COMB[g][w][d-1] = d * 16^w * (g == 0 ? G : base_point2) as 20 field limbs
// (x | y), d = 1..15, w = 0..63 (154 KB, filled once per device by ec_comb_build_kernel when a context is created).
// k*G is then 64 mixed additions and no doublings (jac_mul_comb).
__device__ uint32_t COMB[2][64][15][20];
__global__ void __launch_bounds__(64) MPE_EC_OCC ec_comb_build_kernel() {
  const int w = threadIdx.x & 63, g = blockIdx.x;
  if (g > 1) return;
  Jac b = jac_from_aff(g ? aff_h2() : aff_gen());
  for (int i = 0; i < 4 * w; ++i) b = jac_dbl(b);
  const AffL ba = affl_from_aff(jac_to_aff(b));
  Jac acc = jac_from_affl(ba);
  for (int d = 1; d <= 15; ++d) {
    const AffL a = affl_from_aff(jac_to_aff(acc));
    for (int j = 0; j < 10; ++j) { COMB[g][w][d - 1][j] = a.x.n[j]; COMB[g][w][d - 1][10 + j] = a.y.n[j]; }
    acc = jac_add_affl(acc, ba);
  }
}
// k*G (g = 0) or k*base_point2 (g = 1), k already reduced mod q
__device__ __forceinline__ Jac jac_mul_fixed(const U256& k, int g) { return jac_mul_comb(k, &COMB[g][0][0][0]); }
__device__ __forceinline__ Jac jac_mul_gen(const U256& k) { return jac_mul_fixed(k, 0); }
__device__ __forceinline__ Jac jac_mul_h2(const U256& k) { return jac_mul_fixed(k, 1); }

// interface layout: x[8] | y[8], all-zero = infinity
__device__ __forceinline__ Aff aff_load(const uint32_t* p) {
  Aff a; a.x = u256_load(p); a.y = u256_load(p + 8); a.inf = u256_is_zero(a.x) && u256_is_zero(a.y); return a;
}
__device__ __forceinline__ void aff_store(uint32_t* p, const Aff& a) {
  if (a.inf) { for (int i = 0; i < 16; ++i) p[i] = 0; return; }
  u256_store(p, a.x); u256_store(p + 8, a.y);
}
// A point that arrives from a peer: both coordinates canonical (< p), on the curve y^2 = x^3 + 7, not the point at infinity.
// curv's Point deserialisation performs this check in the reference; without it a secret scalar multiplied into a peer's
// point is open to invalid-curve / small-subgroup inputs.  (secp256k1 has cofactor 1: on-curve = in the group.)
__device__ inline bool aff_valid(const Aff& a) {
  if (a.inf || u256_ge(a.x, FP) || u256_ge(a.y, FP)) return false;
  return aff_on_curve(a);
}
__device__ __forceinline__ bool aff_eq(const Aff& a, const Aff& b) {
  if (a.inf || b.inf) return a.inf && b.inf;
  return u256_eq(a.x, b.x) && u256_eq(a.y, b.y);
}

// ---- SHA-256 (FIPS 180-4), streaming, one hash per lane ------------------------------------------
struct Sha256 {
  uint32_t h[8];
  uint32_t buf[16];      // current block, big-endian words
  uint64_t len;          // bytes absorbed
};
__device__ __constant__ const uint32_t SHA_K[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,
  0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,
  0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,
  0x06ca6351,0x14292967,0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
  0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,0x19a4c116,0x1e376c08,
  0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,
  0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__device__ inline void sha_init(Sha256& s) {
  s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
  s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
  for (int i = 0; i < 16; ++i) s.buf[i] = 0;
  s.len = 0;
}
__device__ __noinline__ void sha_block(Sha256& s) {     // out of line: every sha_byte site would otherwise carry a copy
  uint32_t w[64];
  for (int i = 0; i < 16; ++i) w[i] = s.buf[i];
  for (int i = 16; i < 64; ++i) {
    const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
  for (int i = 0; i < 64; ++i) {
    const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
    const uint32_t t1 = h + S1 + ch + SHA_K[i] + w[i];
    const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
    const uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d; s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
  for (int i = 0; i < 16; ++i) s.buf[i] = 0;
}
__device__ inline void sha_byte(Sha256& s, uint32_t byte) {
  const int off = (int)(s.len & 63);
  s.buf[off >> 2] |= (byte & 0xFFu) << (24 - 8 * (off & 3));
  s.len++;
  if ((s.len & 63) == 0) sha_block(s);
}
// The byte-level conventions of curv / zk-paillier that the reference's own source does not fix are a run-time property of the
// context (mpe_encoding, include/mpecdsa_hip.h): every kernel that hashes takes it by value in its kernel arguments.
using Enc = mpe_encoding;
// DigestExt::chain_bigint: big-endian magnitude, minimal length (0 -> one 0x00 byte, or nothing: enc.zero_bytes)
__device__ inline void sha_bigint(Sha256& s, const uint32_t* x, int nwords, const Enc& enc) {
  int top = nwords - 1;
  while (top > 0 && x[top] == 0) --top;
  int nb = 4;
  const uint32_t tw = x[top];
  if (tw < (1u << 8)) nb = 1; else if (tw < (1u << 16)) nb = 2; else if (tw < (1u << 24)) nb = 3;
  if (top == 0 && tw == 0 && enc.zero_bytes) nb = 0;
  for (int b = nb - 1; b >= 0; --b) sha_byte(s, tw >> (8 * b));
  for (int i = top - 1; i >= 0; --i) {
    const uint32_t w = x[i];
    sha_byte(s, w >> 24); sha_byte(s, w >> 16); sha_byte(s, w >> 8); sha_byte(s, w);
  }
}
// bytes of a field element / coordinate, fixed 32 bytes big-endian
__device__ inline void sha_be32(Sha256& s, const U256& v) {
  for (int i = 7; i >= 0; --i) { const uint32_t w = v.w[i]; sha_byte(s, w >> 24); sha_byte(s, w >> 16); sha_byte(s, w >> 8); sha_byte(s, w); }
}
// Point::to_bytes(true) as hashed by zk_pdl_with_slack (BigInt::from_bytes(33 bytes) -> to_bytes: identical bytes)
__device__ inline void sha_point_compressed(Sha256& s, const Aff& p) { sha_byte(s, 2u + (p.y.w[0] & 1u)); sha_be32(s, p.x); }
// DigestExt::chain_point  [SURVEY.md App. A.2, recalled: the form is enc.chain_point]
__device__ inline void sha_chain_point(Sha256& s, const Aff& p, const Enc& enc) {
  sha_byte(s, enc.chain_point ? 2u + (p.y.w[0] & 1u) : 4u);
  sha_be32(s, p.x);
  if (!enc.chain_point) sha_be32(s, p.y);
}
// digest as 8 little-endian interface words (result_bigint)
__device__ inline U256 sha_final(Sha256& s) {
  const uint64_t bits = s.len * 8;
  sha_byte(s, 0x80);
  while ((s.len & 63) != 56) sha_byte(s, 0);
  for (int i = 7; i >= 0; --i) sha_byte(s, (uint32_t)(bits >> (8 * i)));
  U256 r;
  for (int i = 0; i < 8; ++i) r.w[i] = s.h[7 - i];
  return r;
}

}  // namespace ec
}  // namespace mpe
