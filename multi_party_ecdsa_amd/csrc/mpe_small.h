// "Light" one-item-per-lane multi-precision helpers (plain little-endian u32 words).
//
// Everything that is not a 2048/4096-bit modular multiplication on the GG20 path is glue of
// negligible cost (< 0.5 % of the MACs of one Paillier encryption per item): plain products such as
// s1 = e*a + alpha (range_proofs.rs:88), exact divisions in Paillier's L function, comparisons.
// These run one item per lane with word-serial loops; operands live in global memory or in
// per-lane scratch.  The heavy work stays in mpe_bigint.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpe {
namespace sm {

// r[0..na+nb) = a * b          (r must not alias a or b)
__device__ inline void mul(uint32_t* r, const uint32_t* a, int na, const uint32_t* b, int nb) {
  for (int i = 0; i < na + nb; ++i) r[i] = 0;
  for (int i = 0; i < na; ++i) {
    uint64_t carry = 0;
    const uint64_t ai = a[i];
    for (int j = 0; j < nb; ++j) {
      const uint64_t t = ai * b[j] + r[i + j] + carry;
      r[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    r[i + nb] = (uint32_t)carry;
  }
}
// r[0..n) = low n words of a * b   (a, b have n words; r must not alias)
__device__ inline void mullo(uint32_t* r, const uint32_t* a, const uint32_t* b, int n) {
  for (int i = 0; i < n; ++i) r[i] = 0;
  for (int i = 0; i < n; ++i) {
    uint64_t carry = 0;
    const uint64_t ai = a[i];
    for (int j = 0; i + j < n; ++j) {
      const uint64_t t = ai * b[j] + r[i + j] + carry;
      r[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
  }
}
// r[0..n) = a[0..na) + b[0..nb)  (n >= max(na, nb); returns the carry out of word n-1); r may alias a or b
__device__ inline uint32_t add(uint32_t* r, int n, const uint32_t* a, int na, const uint32_t* b, int nb) {
  uint64_t c = 0;
  for (int i = 0; i < n; ++i) {
    c += (uint64_t)(i < na ? a[i] : 0u) + (i < nb ? b[i] : 0u);
    r[i] = (uint32_t)c;
    c >>= 32;
  }
  return (uint32_t)c;
}
// r[0..n) = a - b (mod 2^(32n)); returns the borrow (1 when a < b); r may alias
__device__ inline uint32_t sub(uint32_t* r, int n, const uint32_t* a, int na, const uint32_t* b, int nb) {
  int64_t c = 0;
  for (int i = 0; i < n; ++i) {
    c += (int64_t)(i < na ? a[i] : 0u) - (int64_t)(i < nb ? b[i] : 0u);
    r[i] = (uint32_t)c;
    c >>= 32;
  }
  return (uint32_t)(c & 1);
}
// compare a[0..na) with b[0..nb): -1, 0, 1
__device__ inline int cmp(const uint32_t* a, int na, const uint32_t* b, int nb) {
  const int n = na > nb ? na : nb;
  for (int i = n - 1; i >= 0; --i) {
    const uint32_t x = i < na ? a[i] : 0u, y = i < nb ? b[i] : 0u;
    if (x != y) return x > y ? 1 : -1;
  }
  return 0;
}
__device__ inline bool is_zero(const uint32_t* a, int n) {
  uint32_t o = 0;
  for (int i = 0; i < n; ++i) o |= a[i];
  return o == 0;
}
__device__ inline void copy(uint32_t* r, const uint32_t* a, int n) {
  for (int i = 0; i < n; ++i) r[i] = a[i];
}
__device__ inline void zero(uint32_t* r, int n) {
  for (int i = 0; i < n; ++i) r[i] = 0;
}
// inv = a^-1 mod 2^(32n) for odd a (Newton / Hensel lifting); t1, t2: n-word scratch
__device__ inline void inv2adic(uint32_t* inv, const uint32_t* a, int n, uint32_t* t1, uint32_t* t2) {
  uint32_t x = a[0];
  for (int i = 0; i < 5; ++i) x *= 2u - a[0] * x;      // 32 correct bits
  zero(inv, n);
  inv[0] = x;
  for (int have = 1; have < n; have *= 2) {            // x <- x * (2 - a*x) doubles the precision
    mullo(t1, a, inv, n);
    // t1 = 2 - t1
    uint64_t c = 2;
    for (int i = 0; i < n; ++i) {
      c += (uint64_t)(uint32_t)~t1[i];
      if (i == 0) c += 1;                              // two's complement: -t1 = ~t1 + 1
      t1[i] = (uint32_t)c;
      c >>= 32;
    }
    mullo(t2, inv, t1, n);
    copy(inv, t2, n);
  }
}

}  // namespace sm
}  // namespace mpe
