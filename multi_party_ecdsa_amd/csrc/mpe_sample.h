// The SAMPLING side of the curv trait surface, on the device (`Samplable::{sample, sample_below, sample_range}`;
// the reference's own `SampleFromMultiplicativeGroup::from_modulo`, src/utilities/mta/range_proofs.rs:538-557; curv's
// `Scalar::<Secp256k1>::random()`).  Everything the signing path draws from OsRng —
//   party_i.rs:561-563 (gamma_i, k_i), :574 (blind factor = BigInt::sample(256)), :628 (l), :620-634, :778-833 (sigma-proof nonces),
//   mta/mod.rs:57 (MessageA randomness), :97-98 (beta_tag, MessageB randomness), :147-148 (DLogProof nonces),
//   mta/range_proofs.rs:48-51 (AliceProof::generate), zk_pdl_with_slack/mod.rs:73-77 (PDLwSlackProof::prove) —
// is produced here from a 32-byte seed by ChaCha20, so a host hands over 32 bytes per batch instead of ~11 KB per session.
//
// The byte -> integer rules are curv's (recalled, curv-kzen 0.9 `arithmetic::traits::Samplable for BigInt`):
//   sample(bits)        = from_bytes_be(ceil(bits/8) fresh bytes) >> (8 ceil(bits/8) - bits)
//   sample_below(u)     = loop { n = sample(bit_length(u)); if n < u return n }
//   sample_range(l, u)  = l + sample_below(u - l)
//   from_modulo(N)      = loop { r = sample_below(N); if gcd(r, N) == 1 return r }            (range_proofs.rs:544-552)
//   Scalar::random()    = loop { 32 fresh bytes as a big-endian integer x; if 0 < x < q return x }   (secp256k1 SecretKey::new)
// so that a generator delivering the same bytes to the reference would make it draw the same values.  What replaces OsRng:
// item g of stream `sid` reads the ChaCha20 keystream (RFC 8439 block function, 20 rounds) with
//   key = seed,  state[12] = block counter from 0,  state[13] = g,  state[14] = low 32 bits of sid,  state[15] = high 32 bits,
// consuming bytes in order; every attempt of a rejection loop takes fresh bytes (rejected draws are part of the stream:
// oracle/sampler_oracle.c expands the same seed to the same arrays, attempts included).
// A (seed, sid) pair must never be used twice — the caller's contract, like a nonce.
// On the GG20 path the ITEM index of a row is the index that row has in the layout with EVERY signer local — ((session * S +
// signer ordinal) * items per party + sub-item) — whatever subset of the signers the calling object hosts: two objects that hold
// different parties of the same batch and are handed the same (seed, batch counter) draw DIFFERENT streams by construction (and an
// object hosting one party draws exactly what the all-local object draws for that party).
// Included by mpe_lib.hip after mpe_gg20.h.
#pragma once
#include "mpe_gg20.h"

namespace mpe {
namespace smp {

// Rejection loops give up after `max_attempts` candidates (mpe_ctx option sampler_max_attempts, default 128: a bound with its top
// bit set rejects with probability < 1/2 per attempt, so an honest draw gives up with probability < 2^-128).  DELIBERATE DIVERGENCE:
// curv's sample_below / from_modulo / Scalar::random loop forever (range_proofs.rs:538-557); a GPU lane must not.
// A draw that gives up writes zeros, counts in *fail, and — on the GG20 path — flags its (session, party): sample_gg20 then makes
// that party's k_i an INVALID scalar (all ones >= q), which round 0 reports as status MPE_GG20_STATUS_BAD_NONCE.  A session never
// signs on zeroed values, whether or not the caller reads *fail.
constexpr int DEFAULT_MAX_ATTEMPTS = 128;
constexpr int F_NONZERO = 1, F_PLUS_ONE = 2, F_COPRIME = 4;
constexpr int NF = 22;                 // fields of mpe_gg20_nonces, msg last

struct Seed { uint32_t k[8]; };

__device__ __forceinline__ uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define MPE_QR(a, b, c, d)                                    \
  a += b; d ^= a; d = rotl(d, 16); c += d; b ^= c; b = rotl(b, 12); \
  a += b; d ^= a; d = rotl(d, 8);  c += d; b ^= c; b = rotl(b, 7);
// one keystream block into out[16] (LDS row of the calling lane)
__device__ inline void chacha20_block(const Seed& key, uint32_t counter, uint32_t n13, uint32_t n14, uint32_t n15, uint32_t* out) {
  uint32_t x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u;
  uint32_t x4 = key.k[0], x5 = key.k[1], x6 = key.k[2], x7 = key.k[3], x8 = key.k[4], x9 = key.k[5], x10 = key.k[6], x11 = key.k[7];
  uint32_t x12 = counter, x13 = n13, x14 = n14, x15 = n15;
  for (int r = 0; r < 10; ++r) {
    MPE_QR(x0, x4, x8, x12) MPE_QR(x1, x5, x9, x13) MPE_QR(x2, x6, x10, x14) MPE_QR(x3, x7, x11, x15)
    MPE_QR(x0, x5, x10, x15) MPE_QR(x1, x6, x11, x12) MPE_QR(x2, x7, x8, x13) MPE_QR(x3, x4, x9, x14)
  }
  out[0] = x0 + 0x61707865u; out[1] = x1 + 0x3320646eu; out[2] = x2 + 0x79622d32u; out[3] = x3 + 0x6b206574u;
  out[4] = x4 + key.k[0]; out[5] = x5 + key.k[1]; out[6] = x6 + key.k[2]; out[7] = x7 + key.k[3];
  out[8] = x8 + key.k[4]; out[9] = x9 + key.k[5]; out[10] = x10 + key.k[6]; out[11] = x11 + key.k[7];
  out[12] = x12 + counter; out[13] = x13 + n13; out[14] = x14 + n14; out[15] = x15 + n15;
}
#undef MPE_QR

// sequential reader of one item's keystream (bytes are asked for in ascending order)
struct Stream {
  const Seed* key; uint32_t n13, n14, n15;
  uint32_t* blk;           // 16 words of LDS owned by this lane
  uint32_t cur;            // block held in blk, or 0xffffffff
  unsigned long long pos;  // next byte
  __device__ __forceinline__ uint32_t next_byte() {
    const uint32_t b = (uint32_t)(pos >> 6);
    if (b != cur) { chacha20_block(*key, b, n13, n14, n15, blk); cur = b; }
    const uint32_t v = (blk[(pos & 63) >> 2] >> (8 * (pos & 3))) & 0xffu;
    ++pos;
    return v;
  }
};

__device__ __forceinline__ int bit_length(const uint32_t* w, int n) {
  for (int j = n - 1; j >= 0; --j) if (w[j]) return 32 * j + 32 - __clz(w[j]);
  return 0;
}

// gcd(a, n) == 1 for ODD n (binary gcd, word-serial; a, n: `w` words, both clobbered)
__device__ inline bool coprime_odd(uint32_t* a, uint32_t* n, int w) {
  if (sm::is_zero(a, w)) return false;
  auto shr_to_odd = [&](uint32_t* x) {
    while (!(x[0] & 1u)) {
      int z = 0;
      while (z < w && x[z] == 0) ++z;                         // whole zero words first
      if (z) { for (int j = 0; j < w; ++j) x[j] = j + z < w ? x[j + z] : 0u; continue; }
      const int s = __ffs(x[0]) - 1;
      for (int j = 0; j < w - 1; ++j) x[j] = (x[j] >> s) | (x[j + 1] << (32 - s));
      x[w - 1] >>= s;
    }
  };
  shr_to_odd(a);
  for (;;) {                                                 // both odd here
    const int c = sm::cmp(a, w, n, w);
    if (c == 0) break;
    if (c > 0) { sm::sub(a, w, a, w, n, w); shr_to_odd(a); }
    else { sm::sub(n, w, n, w, a, w); shr_to_odd(n); }
  }
  uint32_t o = a[0] ^ 1u;
  for (int j = 1; j < w; ++j) o |= a[j];
  return o == 0;
}

// one item: o[0..out_words) drawn below the bound row bd (bits > 0 and bd == nullptr: BigInt::sample(bits), no bound) from the item's stream.
// COPRIME_W: words of the private arrays of the coprimality check (0: the flag is not honoured — the batched verdict is used instead).
template <int COPRIME_W>
__device__ __forceinline__ void draw_item(Stream& s, const uint32_t* __restrict__ bd, int bound_words, int bits, int flags, uint32_t* __restrict__ o,
                                          int out_words, int max_attempts, int32_t* __restrict__ fail, int32_t* __restrict__ owner_bad) {
  const int L = bd ? bit_length(bd, bound_words) : bits;
  if (L <= 0 || L > 32 * out_words) { for (int j = 0; j < out_words; ++j) o[j] = 0; if (fail) atomicAdd(fail, 1); if (owner_bad) *owner_bad = 1; return; }
  const int nbytes = (L + 7) / 8, sh = nbytes * 8 - L, W = (L + 31) / 32;
  bool done = false;
  for (int attempt = 0; attempt < max_attempts && !done; ++attempt) {
    // the integer X = big-endian(bytes) >> sh, L bits.  Bytes arrive most significant first: the first L - 32 (W - 1) bits are
    // little-endian word W - 1 of X, every further 32 bits the next lower word, and the last sh bits of the string are shifted out.
    int state = 0;                       // comparison with the bound so far: 0 equal, -1 below, +1 above
    uint32_t any = 0;
    unsigned long long acc = 0;          // bits read and not yet emitted, most recent in the low positions
    int have = 0, taken = 0;             // bits in acc; bytes read
    for (int j = W - 1; j >= 0; --j) {
      const int need = (j == W - 1 ? L - 32 * (W - 1) : 32);
      while (have < need + (j == 0 ? sh : 0) && taken < nbytes) { acc = (acc << 8) | s.next_byte(); have += 8; ++taken; }
      const int drop = have - need;      // bits that stay for the lower words (j > 0), or the sh bits shifted out (j == 0)
      const uint32_t wv = (uint32_t)((acc >> drop) & (need == 32 ? 0xffffffffull : ((1ull << need) - 1)));
      acc &= drop ? ((1ull << drop) - 1) : 0ull;
      have = drop;
      o[j] = wv;
      any |= wv;
      if (bd && state == 0) { const uint32_t bw = bd[j]; state = wv < bw ? -1 : (wv > bw ? 1 : 0); }
    }
    bool good = bd ? state < 0 : true;
    if (good && (flags & F_NONZERO) && any == 0) good = false;
    if (COPRIME_W > 0) {
      if (good && (flags & F_COPRIME)) {
        uint32_t a[COPRIME_W > 0 ? COPRIME_W : 1], n[COPRIME_W > 0 ? COPRIME_W : 1];
        for (int j = 0; j < COPRIME_W; ++j) { a[j] = j < W ? o[j] : 0u; n[j] = j < bound_words ? bd[j] : 0u; }
        good = (n[0] & 1u) ? coprime_odd(a, n, COPRIME_W) : false;
      }
    }
    done = good;
  }
  for (int j = W; j < out_words; ++j) o[j] = 0;
  if (!done) { for (int j = 0; j < W; ++j) o[j] = 0; if (fail) atomicAdd(fail, 1); if (owner_bad) *owner_bad = 1; return; }
  if (flags & F_PLUS_ONE) { uint32_t c = 1; for (int j = 0; j < out_words && c; ++j) { const uint32_t v = o[j] + c; c = v < c ? 1u : 0u; o[j] = v; } }
}

// row -> stream item of the GG20 arrays (L = 0: the row index itself): rows are [session][local party][per_owner sub-items]
struct ItemMap { int L, S; unsigned per_owner; uint32_t loc_packed; };       // loc_packed: signer ordinal of local party li in bits [4 li, 4 li + 4)
__device__ __forceinline__ uint32_t stream_item(const ItemMap& m, unsigned row) {
  if (m.L == 0) return row;
  const unsigned pil = row / m.per_owner, sub = row % m.per_owner, b = pil / (unsigned)m.L, li = pil % (unsigned)m.L;
  return (b * (unsigned)m.S + ((m.loc_packed >> (4 * li)) & 15u)) * m.per_owner + sub;
}
// out[i] (out_words words, zero-extended) drawn below bound row sel(i); bits > 0: BigInt::sample(bits), no bound.
template <int COPRIME_W>
__global__ void __launch_bounds__(64) sample_kernel(int batch, Seed key, uint32_t sid_lo, uint32_t sid_hi, int bits, const uint32_t* __restrict__ bound,
                                                     int bound_words, const int32_t* __restrict__ bound_idx, int nbounds, int flags, int out_words,
                                                     uint32_t* __restrict__ out, int32_t* __restrict__ fail, const uint8_t* __restrict__ skip_if,
                                                     int max_attempts, int32_t* __restrict__ owner_bad, ItemMap map) {
  __shared__ uint32_t ks[64][17];
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= batch) return;
  if (skip_if && skip_if[i]) return;       // from_modulo, second pass: only the items whose first candidate shares a factor with N
  Stream s{&key, stream_item(map, (unsigned)i), sid_lo, sid_hi, ks[threadIdx.x], 0xffffffffu, 0ull};
  const uint32_t* bd = bound ? bound + (size_t)(bound_idx ? bound_idx[i] : (nbounds == 1 ? 0 : i)) * bound_words : nullptr;
  draw_item<COPRIME_W>(s, bd, bound_words, bits, flags, out + (size_t)i * out_words, out_words, max_attempts, fail, owner_bad ? owner_bad + (unsigned)i / map.per_owner : nullptr);
}

// every field of the sampled values of `nslots` consecutive batches in ONE launch: thread g -> (field, row); a field's rows are
// slot-major (slot = row / rows-per-batch), and slot k draws from ITS batch's (seed, counter): field f of that batch is stream
// counter | f << 56, item = the row index inside the batch.  One slot = mpe_gg20_sample_nonces; several = the pipelined engine's groups.
constexpr int MAX_SLOTS = 16;
struct FieldDesc { uint32_t* out; const uint32_t* bound; const int32_t* idx; unsigned per_slot, per_owner; int bound_words, out_words, bits, flags, field; };
struct FieldTable { FieldDesc f[NF]; unsigned start[NF + 1]; int n; int max_attempts; int32_t* owner_bad; int L, S; uint32_t loc_packed; };    // owner_bad [session-parties]: a draw of theirs gave up
struct SlotTable { Seed key[MAX_SLOTS]; uint32_t ctr_lo[MAX_SLOTS], ctr_hi[MAX_SLOTS]; };
__global__ void __launch_bounds__(64) sample_fields_kernel(FieldTable t, SlotTable sl, int32_t* __restrict__ fail) {
  __shared__ uint32_t ks[64][17];
  __shared__ SlotTable slots;                                // the per-slot keys are indexed per lane: LDS, not kernel-argument SGPRs
  for (unsigned w = threadIdx.x; w < sizeof(SlotTable) / 4; w += 64) ((uint32_t*)&slots)[w] = ((const uint32_t*)&sl)[w];
  __syncthreads();
  const unsigned g = blockIdx.x * 64 + threadIdx.x;
  if (g >= t.start[t.n]) return;
  int f = 0;
  while (g >= t.start[f + 1]) ++f;
  const FieldDesc& d = t.f[f];
  const unsigned row = g - t.start[f], slot = row / d.per_slot, i = row % d.per_slot;
  const ItemMap map{t.L, t.S, d.per_owner, t.loc_packed};
  Stream s{&slots.key[slot], stream_item(map, i), slots.ctr_lo[slot], slots.ctr_hi[slot] | ((uint32_t)d.field << 24), ks[threadIdx.x], 0xffffffffu, 0ull};
  const uint32_t* bd = d.bound ? d.bound + (size_t)(d.idx ? d.idx[row] : 0) * d.bound_words : nullptr;
  draw_item<0>(s, bd, d.bound_words, d.bits, d.flags, d.out + (size_t)row * d.out_words, d.out_words, t.max_attempts, fail,
               t.owner_bad ? t.owner_bad + row / d.per_owner : nullptr);
}
// the party of a session one of whose draws gave up must not sign: its k_i becomes an invalid scalar (>= q), round 0 reports it
__global__ void poison_kernel(int nPI, const int32_t* __restrict__ owner_bad, uint32_t* __restrict__ k) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nPI * 8) return;
  if (owner_bad[g >> 3]) k[g] = 0xFFFFFFFFu;
}

static Seed seed_of(const uint8_t* h) {
  Seed s;
  for (int j = 0; j < 8; ++j) s.k[j] = (uint32_t)h[4 * j] | ((uint32_t)h[4 * j + 1] << 8) | ((uint32_t)h[4 * j + 2] << 16) | ((uint32_t)h[4 * j + 3] << 24);
  return s;
}

static int launch_sample(int batch, const uint8_t* h_seed, uint64_t sid, int bits, const uint32_t* d_bound, int bound_words, const int32_t* d_bound_idx,
                         int nbounds, int flags, int out_words, uint32_t* d_out, int32_t* d_fail, hipStream_t st, const uint8_t* d_skip_if = nullptr,
                         int max_attempts = DEFAULT_MAX_ATTEMPTS, int32_t* d_owner_bad = nullptr, ItemMap map = ItemMap{0, 0, 1u, 0u}) {
  if (batch == 0) return MPE_OK;
  if (!h_seed || !d_out || out_words <= 0 || (d_bound ? (bound_words <= 0 || bound_words > out_words || nbounds < 1) : (bits <= 0 || bits > 32 * out_words)))
    return MPE_E_ARG;
  if ((flags & F_COPRIME) && (!d_bound || bound_words > 64)) return MPE_E_ARG;
  const Seed key = seed_of(h_seed);
  const dim3 grid(blocks_for(batch, 64)), block(64);
  if (flags & F_COPRIME)
    hipLaunchKernelGGL(sample_kernel<64>, grid, block, 0, st, batch, key, (uint32_t)sid, (uint32_t)(sid >> 32), bits, d_bound, bound_words, d_bound_idx,
                       nbounds, flags, out_words, d_out, d_fail, d_skip_if, max_attempts, d_owner_bad, map);
  else
    hipLaunchKernelGGL(sample_kernel<0>, grid, block, 0, st, batch, key, (uint32_t)sid, (uint32_t)(sid >> 32), bits, d_bound, bound_words, d_bound_idx,
                       nbounds, flags, out_words, d_out, d_fail, d_skip_if, max_attempts, d_owner_bad, map);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("sample_kernel", e); return MPE_E_HIP; }
  return MPE_OK;
}

// which bound row every item of a nonce field uses (same index conventions as mpe_gg20.h)
struct SIdx { int32_t *own_pi, *own_ap, *st_ap, *peer_mb, *own_pp, *st_pp; };
__global__ void sidx_kernel(gg::Dim d, SIdx x, int total) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const int n = d.n, L = d.L, P1 = d.S - 1;
  if (g < d.B * L) x.own_pi[g] = gg::kpub(d, g / L, d.loc[g % L]);
  if (g < d.B * L * n) { const int pi = g / n; x.own_ap[g] = gg::kpub(d, pi / L, d.loc[pi % L]); x.st_ap[g] = gg::ks_of(d, pi / L) * n + g % n; }
  if (g < d.B * L * P1 * 2) { const int pp = g >> 1, pi = pp / P1; x.peer_mb[g] = gg::kpub(d, pi / L, gg::ind_of(d.loc[pi % L], pp % P1)); }
  if (g < d.B * L * P1) {
    const int pi = g / P1, b = pi / L, i = d.loc[pi % L];
    x.own_pp[g] = gg::kpub(d, b, i); x.st_pp[g] = gg::ks_of(d, b) * n + d.sg[gg::ind_of(i, g % P1)];
  }
}

// words per session-party of every field of mpe_gg20_nonces, in declaration order (msg last: per session)
static void nonce_field_words(int S, int n, int L, size_t per_session[NF]) {
  const size_t P1 = (size_t)S - 1, l = (size_t)L;
  const size_t w[NF] = {l * 8, l * 8, l * 8, l * 64, l * n * 24, l * n * 64, l * n * 88, l * n * 72, l * P1 * 2 * 64, l * P1 * 2 * 64, l * P1 * 2 * 8, l * P1 * 2 * 8,
                        l * 8, l * 8, l * 8, l * P1 * 24, l * P1 * 64, l * P1 * 72, l * P1 * 88, l * 8, l * 8, 8};
  for (int f = 0; f < NF; ++f) per_session[f] = w[f];
}
static const uint32_t** nonce_field_ptr(mpe_gg20_nonces* z, int f) {
  const uint32_t** p[NF] = {&z->k, &z->gamma, &z->blind, &z->r_a, &z->al_alpha, &z->al_beta, &z->al_gamma, &z->al_rho, &z->mb_beta_tag, &z->mb_r, &z->mb_nonce_b,
                            &z->mb_nonce_bt, &z->l, &z->ped_s1, &z->ped_s2, &z->pdl_alpha, &z->pdl_beta, &z->pdl_rho, &z->pdl_gamma, &z->heg_s1, &z->heg_s2, &z->msg};
  return p[f];
}

}  // namespace smp
}  // namespace mpe

// device arrays for the sampled values of `batch` sessions x n_local parties (one allocation, wiped when released)
struct mpe_gg20_nonce_buf {
  void* blob = nullptr;
  size_t bytes = 0;
  int S = 0, n = 0, L = 0, batch = 0;
  mpe_gg20_nonces view{};
  size_t per_session[mpe::smp::NF] = {0};
};

namespace mpe {
namespace smp {

// `nslots` consecutive batches of `per_batch` sessions each (arrays slot-major), slot k drawn from (h_seeds + 32 k, h_counters[k])
static int sample_gg20(mpe_ctx* ctx, const mpe_gg20_keys* K, int per_batch, int nslots, int n_local, const int32_t* h_local, const int32_t* d_keyset,
                       const uint8_t* h_seeds, const uint64_t* h_counters, const mpe_gg20_nonces* out, int32_t* d_fail, hipStream_t st) {
  if (!ctx || !K || !h_local || !h_seeds || !h_counters || !out || per_batch < 0 || nslots < 1 || nslots > MAX_SLOTS || n_local < 1 || n_local > K->S) return MPE_E_ARG;
  for (int k = 0; k < nslots; ++k) if ((h_counters[k] >> 56) != 0) return MPE_E_ARG;
  if (K->K > 1 && !d_keyset) return MPE_E_ARG;
  if (per_batch == 0) return MPE_OK;
  const int batch = per_batch * nslots;
  gg::Dim d{};
  d.B = batch; d.S = K->S; d.n = K->n; d.L = n_local; d.K = K->K; d.n_own = K->n_own; d.ks = d_keyset;
  for (int i = 0; i < 8; ++i) { d.loc[i] = i < n_local ? h_local[i] : 0; d.sg[i] = K->signers[i]; d.oslot[i] = K->own_slot[i] < 0 ? 0 : K->own_slot[i]; }
  for (int i = 0; i < n_local; ++i) if (h_local[i] < 0 || h_local[i] >= K->S || (i && h_local[i] <= h_local[i - 1])) return MPE_E_ARG;
  const size_t P1 = (size_t)K->S - 1, nPI = (size_t)batch * n_local, nAP = nPI * K->n, nPP = nPI * P1, nMB = nPP * 2;
  // workspace: the bound-row index of every item + the batched coprimality verdict of from_modulo
  const mpe_modset* msn = K->pub->ms_n;
  const size_t idx_words = 2 * nPI + 2 * nAP + nMB + 2 * nPP + 64;
  const size_t inv_bytes = (modinv_ws_words(msn, (int)nAP) + nAP * 64) * 4 + nAP + 4096;
  MPE_TRY(ws_reserve(ctx, idx_words * 4 + inv_bytes + 16 * 256, st));
  SIdx x{ws_array<int32_t>(ctx, nPI), ws_array<int32_t>(ctx, nAP), ws_array<int32_t>(ctx, nAP), ws_array<int32_t>(ctx, nMB), ws_array<int32_t>(ctx, nPP),
         ws_array<int32_t>(ctx, nPP)};
  uint32_t* inv = ws_array<uint32_t>(ctx, nAP * 64);
  uint8_t* ok = ws_array<uint8_t>(ctx, nAP);
  int32_t* owner_bad = ws_array<int32_t>(ctx, nPI);           // a draw of this (session, party) gave up
  if (!x.own_pi || !x.own_ap || !x.st_ap || !x.peer_mb || !x.own_pp || !x.st_pp || !inv || !ok || !owner_bad) return MPE_E_NOMEM;
  (void)hipMemsetAsync(owner_bad, 0, nPI * 4, st);
  size_t total = nMB > nAP ? nMB : nAP;
  if (total < nPI) total = nPI;
  hipLaunchKernelGGL(sidx_kernel, dim3(blocks_for((int)total, 64)), dim3(64), 0, st, d, x, (int)total);
  const Bounds& b = K->bounds;
  const int nk = K->K * K->n;
  const uint32_t* N = K->pub->N;
  auto U = [](const uint32_t* p) { return const_cast<uint32_t*>(p); };
  // field ids = the position in mpe_gg20_nonces (include/mpecdsa_hip.h); all fields in ONE launch
  FieldTable t{};
  t.n = 0; t.start[0] = 0; t.max_attempts = ctx->sampler_max_attempts; t.owner_bad = owner_bad;
  t.L = n_local; t.S = K->S; t.loc_packed = 0;
  for (int i = 0; i < n_local; ++i) t.loc_packed |= (uint32_t)h_local[i] << (4 * i);
  auto add = [&](int f, const uint32_t* dst, size_t items, const uint32_t* bound, int bw, const int32_t* idx, int ow, int flags, int bits = 0) {
    t.f[t.n] = FieldDesc{U(dst), bound, idx, (unsigned)(items / nslots), (unsigned)(items / nPI), bw, ow, bits, flags, f};
    t.start[t.n + 1] = t.start[t.n] + (unsigned)items;
    t.n++;
  };
  auto scalar = [&](int f, const uint32_t* dst, size_t items) { add(f, dst, items, b.q, 8, nullptr, 8, F_NONZERO); };      // Scalar::random()
  scalar(0, out->k, nPI);                                                                      // party_i.rs:563 k_i
  scalar(1, out->gamma, nPI);                                                                  // :561 gamma_i
  add(2, out->blind, nPI, nullptr, 0, nullptr, 8, 0, 256);                                     // :574 BigInt::sample(SECURITY)
  add(3, out->r_a, nPI, N, 64, x.own_pi, 64, 0);                                               // mta/mod.rs:57 sample_below(&alice_ek.n)
  add(4, out->al_alpha, nAP, b.q3, 24, nullptr, 24, 0);                                        // range_proofs.rs:48 sample_below(q^3)
  add(5, out->al_beta, nAP, N, 64, x.own_ap, 64, 0);                                           // :49 from_paillier_key -> from_modulo (:544-552)
  add(6, out->al_gamma, nAP, b.q3Nt, 88, x.st_ap, 88, 0);                                      // :50 sample_below(q^3 N~)
  add(7, out->al_rho, nAP, b.qNt, 72, x.st_ap, 72, 0);                                         // :51 sample_below(q N~)
  add(8, out->mb_beta_tag, nMB, N, 64, x.peer_mb, 64, 0);                                      // mta/mod.rs:97 sample_below(&alice_ek.n)
  add(9, out->mb_r, nMB, N, 64, x.peer_mb, 64, 0);                                             // :98
  scalar(10, out->mb_nonce_b, nMB);                                                            // :147 DLogProof::prove(b)
  scalar(11, out->mb_nonce_bt, nMB);                                                           // :148
  scalar(12, out->l, nPI);                                                                     // party_i.rs:628 l
  scalar(13, out->ped_s1, nPI);                                                                // PedersenProof::prove (:620-634)
  scalar(14, out->ped_s2, nPI);
  add(15, out->pdl_alpha, nPP, b.q3, 24, nullptr, 24, 0);                                      // zk_pdl_with_slack/mod.rs:73
  add(16, out->pdl_beta, nPP, b.Nm2, 64, x.own_pp, 64, F_PLUS_ONE);                            // :75 sample_range(1, N - 1)
  add(17, out->pdl_rho, nPP, b.qNt, 72, x.st_pp, 72, 0);                                       // :76
  add(18, out->pdl_gamma, nPP, b.q3Nt, 88, x.st_pp, 88, 0);                                    // :77
  scalar(19, out->heg_s1, nPI);                                                                // HomoELGamalProof::prove (:778-799)
  scalar(20, out->heg_s2, nPI);
  if ((size_t)t.start[t.n] != 9 * nPI + 4 * nAP + 4 * nMB + 4 * nPP) { mpe_set_error_msg("gg20 sampler: more than 2^32 items in one batch"); return MPE_E_ARG; }          // a batch beyond 2^32 items
  SlotTable sl{};
  for (int k = 0; k < nslots; ++k) { sl.key[k] = seed_of(h_seeds + 32 * k); sl.ctr_lo[k] = (uint32_t)h_counters[k]; sl.ctr_hi[k] = (uint32_t)(h_counters[k] >> 32); }
  hipLaunchKernelGGL(sample_fields_kernel, dim3((t.start[t.n] + 63) / 64), dim3(64), 0, st, t, sl, d_fail);
  {  // from_modulo: gcd(r, N) == 1 as ONE batched verdict for all items (Montgomery's trick, mpe_modinv.h); whoever fails it — 2^-1023
     // per draw for an honest key — is redrawn by the lane-serial loop, which replays the item's stream with the gcd inside the loop
    MPE_TRY(launch_modinv(ctx, msn, (int)nAP, key_selector(K->pub, x.own_ap), rows(out->al_beta, 64), inv, ok, st));
    const size_t per = nAP / nslots;              // the redraw pass runs slot by slot: every slot has its own stream
    for (int k = 0; k < nslots; ++k)
      MPE_TRY(launch_sample((int)per, h_seeds + 32 * k, h_counters[k] | ((uint64_t)5 << 56), 0, N, 64, x.own_ap + k * per, nk, F_COPRIME, 64,
                            U(out->al_beta) + k * per * 64, d_fail, st, ok + k * per, ctx->sampler_max_attempts, owner_bad + k * (per / K->n),
                            ItemMap{n_local, K->S, (unsigned)K->n, t.loc_packed}));
  }
  hipLaunchKernelGGL(poison_kernel, dim3(blocks_for((int)nPI * 8, 256)), dim3(256), 0, st, (int)nPI, owner_bad, U(out->k));
  // the verdict array and the discarded inverses are derived from secret values
  (void)hipMemsetAsync(inv, 0, nAP * 64 * 4, st);
  (void)hipMemsetAsync(ok, 0, nAP, st);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("gg20 sampler", e); return MPE_E_HIP; }
  return MPE_OK;
}

}  // namespace smp
}  // namespace mpe

extern "C" {

int mpe_sample_bits(mpe_ctx* ctx, int batch, const uint8_t* h_seed32, uint64_t stream_id, int bits, int out_words, uint32_t* d_out, void* stream) {
  if (!ctx) return MPE_E_ARG;
  return mpe::smp::launch_sample(batch, h_seed32, stream_id, bits, nullptr, 0, nullptr, 0, 0, out_words, d_out, nullptr, (hipStream_t)stream, nullptr, ctx->sampler_max_attempts);
}
int mpe_sample_below(mpe_ctx* ctx, int batch, const uint8_t* h_seed32, uint64_t stream_id, const uint32_t* d_bound, int bound_words, int nbounds,
                     const int32_t* d_bound_idx, int flags, int out_words, uint32_t* d_out, int32_t* d_fail, void* stream) {
  if (!ctx || !d_bound || (flags & ~7)) return MPE_E_ARG;
  return mpe::smp::launch_sample(batch, h_seed32, stream_id, 0, d_bound, bound_words, d_bound_idx, nbounds, flags, out_words, d_out, d_fail, (hipStream_t)stream, nullptr,
                                 ctx->sampler_max_attempts);
}
int mpe_sample_scalar(mpe_ctx* ctx, int batch, const uint8_t* h_seed32, uint64_t stream_id, uint32_t* d_out, int32_t* d_fail, void* stream) {
  if (!ctx) return MPE_E_ARG;
  static const uint32_t q[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  MPE_TRY(mpe::ws_reserve(ctx, 4096, (hipStream_t)stream));
  uint32_t* dq = mpe::ws_array<uint32_t>(ctx, 8);
  if (!dq) return MPE_E_NOMEM;
  (void)hipMemcpyAsync(dq, q, 32, hipMemcpyHostToDevice, (hipStream_t)stream);
  return mpe::smp::launch_sample(batch, h_seed32, stream_id, 0, dq, 8, nullptr, 1, mpe::smp::F_NONZERO, 8, d_out, d_fail, (hipStream_t)stream, nullptr,
                                 ctx->sampler_max_attempts);
}

int mpe_gg20_nonces_alloc(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, int n_local, mpe_gg20_nonce_buf** out) {
  if (!ctx || !keys || !out || batch <= 0 || n_local < 1 || n_local > keys->S) return MPE_E_ARG;
  mpe_gg20_nonce_buf* nb = new (std::nothrow) mpe_gg20_nonce_buf();
  if (!nb) return MPE_E_NOMEM;
  nb->S = keys->S; nb->n = keys->n; nb->L = n_local; nb->batch = batch;
  mpe::smp::nonce_field_words(keys->S, keys->n, n_local, nb->per_session);
  size_t words = 0;
  for (int f = 0; f < mpe::smp::NF; ++f) words += ((nb->per_session[f] * batch + 63) & ~(size_t)63);
  nb->bytes = words * 4;
  const hipError_t e = hipMalloc(&nb->blob, nb->bytes);
  if (e != hipSuccess) { delete nb; mpe_set_error("hipMalloc(gg20 nonces)", e); return MPE_E_NOMEM; }
  uint32_t* p = (uint32_t*)nb->blob;
  for (int f = 0; f < mpe::smp::NF; ++f) { *mpe::smp::nonce_field_ptr(&nb->view, f) = p; p += ((nb->per_session[f] * batch + 63) & ~(size_t)63); }
  (void)hipMemset(nb->blob, 0, nb->bytes);
  *out = nb;
  return MPE_OK;
}
int mpe_gg20_nonces_view(const mpe_gg20_nonce_buf* nb, mpe_gg20_nonces* out) {
  if (!nb || !out) return MPE_E_ARG;
  *out = nb->view;
  return MPE_OK;
}
int mpe_gg20_nonces_free(mpe_gg20_nonce_buf* nb) {
  if (!nb) return MPE_E_ARG;
  if (nb->blob) { (void)hipMemset(nb->blob, 0, nb->bytes); (void)hipFree(nb->blob); }      // k_i, gamma_i, Paillier randomness: wiped
  delete nb;
  return MPE_OK;
}
int mpe_gg20_sample_nonces(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, int n_local, const int32_t* h_local, const int32_t* d_keyset,
                           const uint8_t* h_seed32, uint64_t batch_counter, const mpe_gg20_nonces* out, int32_t* d_fail, void* stream) {
  return mpe::smp::sample_gg20(ctx, keys, batch, 1, n_local, h_local, d_keyset, h_seed32, &batch_counter, out, d_fail, (hipStream_t)stream);
}

}  // extern "C"
