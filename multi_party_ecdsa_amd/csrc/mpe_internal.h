// Internal definitions shared by the parts of libmpecdsa_hip.so (one translation unit: mpe_lib.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <new>
#include <utility>
#include <string>
#include <vector>

#include "../../include/mpecdsa_hip.h"
#include "mpe_kernels_heavy.h"
#include "mpe_sched.h"

struct mpe_ctx {
  int device = 0;
  int cus = 256;
  int fb_window_bits = 13;        // window width of the fixed-base tables of h1, h2 (mpe_fixedbase.h); 0.5 GB per base at 13
  int window_bits = 0;            // 0 = choose per exponent length (4/5/6); 4..6 = force (A/B runs)
  bool ec_lane_groups = true;     // small batches: a group of lanes per party in the EC round kernels (mpe_gg20.h)
  bool adaptive_lanes = true;     // small batches: twice the lanes per exponentiation (mpe_pair*.hip)
  bool use_pown = true;           // key holders: x^N mod p^2 as (x^(q mod (p-1)) mod p)^p (mpe_paillier.h modexp_nn)
  bool use_pair = true;           // arithmetic modulo N^2 / p^2 in N-adic pair form (mpe_pairexp.h): half the multiplies
  bool use_multiexp = true;       // verifiers: s^N * (c^-1)^e on one ladder instead of two exponentiations (same residue)
  bool use_crt = true;            // key holders compute x^e mod N^2 through p^2 | q^2 (mpe_paillier.h modexp_nn)
  bool use_fixed_base = true;     // h1/h2 exponentiations through per-statement window tables (mpe_fixedbase.h)
  bool use_sliding = true;        // x^N with the PUBLIC exponent N: items ordered by key, sliding windows per wave (mpe_pairexp.h)
  int wide_div = 4;               // the 2x-lanes layout is used when wide_div * batch <= the resident groups: 4 = while its units fit ONE wave per
                                  // SIMD (lone 9-limb ladder 24 ms); up to twice that the 18-limb layout still runs lone (35 ms) where 9 limbs
                                  // would put two waves on every SIMD (38 ms) — re-measured with the scheduler of round 6 (rounds 2-5: 2)
  int device_share = 1;           // contexts expected to run on this device AT THE SAME TIME (mpe_ctx_set_device_share): the small-batch
                                  // heuristics below compare a launch with 1/device_share of the chip, not with all of it
  int xwide_div = 16;             // the 4x-lanes (5 limbs per lane) layout: xwide_div * batch <= the resident groups; 0 = off (MPE_XWIDE_DIV)
  bool merge_xn = true;           // round 0: every x^N of the key holders in ONE launch (MPE_NO_MERGE_XN switches it off)
  int use_prio = 1;               // wave priorities in the small-batch schedule (option no_prio; the pipelined engine's lanes run without)
  int ladder_prio = 1;            // s_setprio of the NEXT ladder launches (mpe_sched.h wave_priority): 1 = the default of ladders, 2 = the pair
                                  // engine's launches (the stretches a small batch waits for), 0 = work started ahead of its round
  int use_crt_n = 1;              // the provers' r^e mod N through p | q (mpe_paillier.h modexp_n_holder; option no_crt_n)
  int no_pdl_ahead = 0;           // lock-step signing of small batches: round 4 computes the PDL proofs' beta^N itself (option)
  int no_r1_dlog_first = 0;       // with the inversion ahead: MessageB's DLog proofs behind the ladders again, not in front of the N~ side (option)
  int no_r1_inversion_ahead = 0;  // lock-step signing of small batches: round 1 inverts the ciphertexts itself, as the per-round calls do (option)
  int merge_r1_quarters = 1;      // small batches merge round 1's two ladder launches when together they exceed this many QUARTERS of the resident groups (option)
  bool merge_r1 = true;           // round 1, large batches: the ladders of the verifications and of the MessageBs in ONE launch (MPE_NO_MERGE_R1)
  size_t fb_budget_bytes = 0;     // memory budget of the fixed-base tables of a key object; 0 = a quarter of free HBM (MPE_FB_BUDGET_MB)
  int modexp_waves_per_cu = 8;    // 2 waves/SIMD: the montmul loop holds ~230 VGPRs and already issues back-to-back
  int fb_split = 0;               // fixed-base ladders: lane groups that share one item's windows (0 = chosen per launch, mpe_fixedbase.h)
  int gg20_trace = 0;             // option gg20_trace: synchronise and report after every composite of a round
  int sampler_max_attempts = 128; // rejection loops of the device sampler give up after this many candidates (mpe_sample.h)
  int no_primaries = 0;           // option no_primaries: launches with at most one unit per SIMD keep static units (the queue of multi-pass launches stays)
  int no_elect = 0;               // option no_elect: the dispatcher's placement is taken as it comes (rounds 1-5), mpe_sched.h
  int grid_mode = 2;              // persistent_grid(): 0 = equal trips (rounds 1-4), 1 = full trips + tail, 2 = tail only when it fits one wave per SIMD (option grid = equal|full|hybrid)
  // window-table scratch, grown on demand: one buffer per stream slot (0 = the caller's stream, 1..3 = the auxiliary streams
  // on which small batches run independent launches concurrently)
  void* tables[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // slot 0: the caller's stream, 1..3: the auxiliary streams
  size_t tables_bytes[6] = {0, 0, 0, 0, 0, 0};
  // Small batches are latency-bound (a launch lasts as long as ONE exponentiation): independent parts of a proof / a round
  // run on auxiliary streams, forked from and joined to the caller's stream with events (mpe::Fork).
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_ahead = nullptr;  // "the PDL proofs' beta^N, started in round 2, is done" (mpe_gg20.h round2 / round4)
  hipEvent_t ev_mid = nullptr;    // "the merged ladder launch of round 1 is queued" (mpe_gg20.h round1)
  // background streams of the lock-step composition (small batches): the pure verifications of rounds 1 and 5 run there,
  // each with its own workspace, and are joined when the signature is completed
  bool aux_ready = false;
  bool allow_par = true;          // MPE_NO_PAR=1 switches the concurrency off (A/B runs)
  int par_items = 32768;          // composites fork when they have at most this many items
  // bump-allocated workspace for the intermediates of composite operations (Paillier, proofs)
  void* ws = nullptr;
  size_t ws_bytes = 0, ws_off = 0;
  int ws_hold = 0;                // > 0: a caller runs several composites concurrently out of ONE reservation: ws_reserve only checks
  size_t ws_top = 0;              // bytes a composite keeps at the TOP of the workspace (its own arrays): never handed out, never moved
  // cached device memory of the GG20 round pipeline: one session object's state (mpe_gg20_session) and the message slabs of
  // mpe_gg20_sign — kept across calls so that a step does not pay hipMalloc / hipFree
  void* sess_buf = nullptr;
  size_t sess_bytes = 0;
  bool sess_in_use = false;
  void* slab_buf = nullptr;
  size_t slab_bytes = 0;
  mpe_launch_info last = {};
  // byte-level conventions of the un-vendored crates (include/mpecdsa_hip.h: mpe_encoding); kernels take it by value
  mpe_encoding enc = {};
  // optional per-launch timing of the heavy kernels (HIP events on the launch stream)
  bool prof_on = false;
  struct ProfEvt { hipEvent_t a, b; int kind, bits, exp_words, batch, exp2_words; int wave_trips; };
  std::vector<ProfEvt> prof;
  // one device counter per record (allocated by mpe_prof_enable): the pair kernel adds 1 per wave-trip that really ran the
  // sliding-window schedule — "kind 6" is a property of the launch, whether a wave slides is decided at run time
  uint32_t* prof_ctr = nullptr;
  int prof_ctr_cap = 0;
};

struct mpe_modset {
  int bits = 0;
  int count = 0;
  int K = 0;
  void* blob = nullptr;
  uint32_t* n_limbs = nullptr;    // [count][K]   modulus, internal radix
  uint32_t* one_limbs = nullptr;  // [count][K]   R mod n
  uint32_t* r2_limbs = nullptr;   // [count][K]   R^2 mod n
  uint32_t* r2h_limbs = nullptr;  // [count][K]   2^bits R^2 mod n
  uint32_t* n0inv = nullptr;      // [count]      -n^-1 mod 2^W
  uint32_t* words = nullptr;      // [count][bits/32] the moduli as interface words (for modinv)
  uint32_t* one_words = nullptr;  // [1]          the constant 1 (a 1-word operand: x * 1 mod n reduces x)
};

// per-modulus constants of the N-adic pair arithmetic modulo N^2 (mpe_pairexp.h)
struct mpe_pairset {
  int half_bits = 0;     // bits of the modulus N (the arithmetic is modulo N^2)
  int count = 0;
  void* blob = nullptr;
  uint32_t *n_limbs = nullptr, *n0inv = nullptr, *one = nullptr, *r2 = nullptr, *tp = nullptr, *kc = nullptr;
  const uint32_t* mod_words = nullptr;   // [count][half_bits/32], owned by the caller (the key set)
};

void mpe_set_error(const char* what, hipError_t e);
void mpe_set_error_msg(const char* what);

namespace mpe {

inline Rows rows(const uint32_t* p, int stride, const int32_t* idx = nullptr, int words = 0) {
  return Rows{p, idx, stride, words};
}
inline Rows no_rows() { return Rows{nullptr, nullptr, 0, 0}; }

// mod_sel: idx != null -> idx[i];  stride != 0 -> i;  else modulus 0
int launch_modexp(mpe_ctx* ctx, const mpe_modset* ms, int batch, Rows mod_sel, Rows base_lo, Rows base_hi, Rows exps,
                  int exp_words, uint32_t* out, hipStream_t st);
// base^exps * base2^exps2 on one ladder (shared squarings); exps2 short (32 exp2_words < 32 exp_words)
int launch_modexp2(mpe_ctx* ctx, const mpe_modset* ms, int batch, Rows mod_sel, Rows base, Rows exps, int exp_words,
                   Rows base2, Rows exps2, int exp2_words, uint32_t* out, hipStream_t st);
int launch_modmul(mpe_ctx* ctx, const mpe_modset* ms, int batch, Rows mod_sel, Rows a, Rows b, uint32_t* out,
                  hipStream_t st);
int modset_create_dev(mpe_ctx* ctx, int bits, int count, const uint32_t* d_moduli, mpe_modset** out, hipStream_t st);

// dst takes every run-time option (mpe_ctx_set_option) of src; defined beside the option table in mpe_lib.hip
void ctx_copy_options(mpe_ctx* dst, const mpe_ctx* src);
// per-launch timing records (mpe_prof_*), defined in mpe_lib.hip
void prof_begin(mpe_ctx* ctx, hipStream_t st, int kind, int bits, int exp_words, int batch, int exp2_words = 0);
void prof_end(mpe_ctx* ctx, hipStream_t st);
// device counter of the record prof_begin just opened (nullptr when profiling is off or the counters are exhausted);
// wave_trips = the number of (wave, trip) pairs of the launch, the counter's maximum
uint32_t* prof_counter(mpe_ctx* ctx, int wave_trips);

// N-adic pair engine (mpe_pair2048.hip / mpe_pair1024.hip)
int pairset_create_2048(int count, const uint32_t* d_moduli, mpe_pairset** out, hipStream_t st);
int pairset_create_1024(int count, const uint32_t* d_moduli, mpe_pairset** out, hipStream_t st);
int pair_modexp_2048(mpe_ctx* ctx, const mpe_pairset* ps, int batch, Rows mod_sel, Rows base, Rows exps, int exp_words,
                     Rows base2, Rows exps2, int exp2_words, int half, uint32_t* out, hipStream_t st, int public_exp);
int pair_modexp_1024(mpe_ctx* ctx, const mpe_pairset* ps, int batch, Rows mod_sel, Rows base, Rows exps, int exp_words,
                     Rows base2, Rows exps2, int exp2_words, int half, uint32_t* out, hipStream_t st, int public_exp);

// workspace: reserve once per composite call (may reallocate -> synchronises the stream), then bump-allocate
int ws_reserve(mpe_ctx* ctx, size_t bytes, hipStream_t st);
void* ws_alloc(mpe_ctx* ctx, size_t bytes);
// window-table scratch of the stream slot `st` belongs to (nullptr + error set when it cannot be grown)
uint32_t* tables_for(mpe_ctx* ctx, size_t need, hipStream_t st);
bool ensure_aux(mpe_ctx* ctx);

// fork / join of up to 3 concurrent branches: branch 0 stays on the caller's stream, branch i > 0 runs on an auxiliary stream.
// `first`: which auxiliary streams (composite-internal forks use 0, 1; the round-level fork uses 2 so that the two nest).
struct Fork {
  mpe_ctx* ctx; hipStream_t main; int n, first; bool on;
  Fork(mpe_ctx* c, hipStream_t m, int branches, bool enable, int first_aux = 0);
  hipStream_t s(int i) const { return (on && i > 0) ? ctx->aux[first + i - 1] : main; }
  void join();
  void branch_done_wait(int i, hipStream_t waiter);     // `waiter` waits for what branch i has queued so far
};
// a composite's own arrays at the top of the workspace: reserved against ws_alloc / ws_reserve until the composite returns
struct WsTop {
  mpe_ctx* c;
  WsTop(mpe_ctx* ctx, const char* top) : c(ctx) { c->ws_top = (size_t)(((const char*)c->ws + c->ws_bytes) - top); }
  ~WsTop() { c->ws_top = 0; }
};
template <class T>
inline T* ws_array(mpe_ctx* ctx, size_t count) { return (T*)ws_alloc(ctx, count * sizeof(T)); }

inline int blocks_for(int n, int threads) { return (n + threads - 1) / threads; }

// Persistent grid of the ladder kernels: `units` wave-units of work on `cap` resident wave slots (2 per SIMD).  What the per-wave
// trace says (mpe_sched.h): a SIMD favours its older wave, two waves deliver 1.19x the units of one, and a launch lasts as long as
// its slowest wave.  Hence
//   units <= cap / 2          up to 2 x units workgroups start, the first arrival on every SIMD is its PRIMARY and the primaries take the
//                             units from a queue: never two units of the launch side by side on one SIMD, whatever the dispatcher does;
//   cap / 2 < units <= cap    one workgroup per unit, static (most SIMDs hold two either way);
//   units > cap               cap workgroups, EVERY unit from the queue: both waves of a SIMD stay busy until the queue is dry (with
//                             static units the favoured wave left early and the other ran on alone), tails balance themselves.
// Option no_elect = 1 restores round 5: static units, `grid` = equal | full | hybrid deciding how a launch of n.f passes is cut
// (A/B files: profiles/r05/ab_grid_three_modes.jsonl, profiles/r06/).
inline int persistent_grid(const mpe_ctx* ctx, int need, int cap);
inline int ladder_grid(const mpe_ctx* ctx, int units, int cap) {
  if (ctx->no_elect) return persistent_grid(ctx, units, cap);
  if (units <= cap) return (2 * units <= cap && !ctx->no_primaries) ? 2 * units : units;
  return cap;
}
// the scheduler arguments of that launch; `state` (SCHED_WORDS ints of device scratch the launch owns) is zeroed on `st` when used
inline SchedArgs ladder_sched(const mpe_ctx* ctx, int units, int cap, int32_t* state, hipStream_t st) {
  SchedArgs a{nullptr, SCHED_STATIC, units, ctx->use_prio ? ctx->ladder_prio : 0};
  if (ctx->no_elect || !state) return a;
  if (2 * units <= cap) { if (!ctx->no_primaries) { a.state = state; a.mode = SCHED_PRIMARIES; } }
  else if (units > cap) { a.state = state; a.mode = SCHED_ALL; }
  if (a.state) (void)hipMemsetAsync(a.state, 0, SCHED_WORDS * sizeof(int32_t), st);
  return a;
}
// kernels WITHOUT the scheduler (modmul, the inversion sweeps): static units, the low block indices own the tail
inline int persistent_grid(const mpe_ctx* ctx, int need, int cap) {
  if (need <= cap) return need;
  const int rem = need % cap;
  if (rem == 0) return cap;
  if (ctx->grid_mode == 1 || (ctx->grid_mode == 2 && 2 * rem <= cap)) return cap;
  const int trips = (need + cap - 1) / cap;
  return (need + trips - 1) / trips;
}

}  // namespace mpe
