// libmpecdsa_hip.so — C-ABI (include/mpecdsa_hip.h) over the gfx950 kernels.  Single translation unit.
#include <string.h>
#include <initializer_list>
#include <mutex>
#include "mpe_internal.h"
#include "mpe_ec.h"

using namespace mpe;

thread_local std::string g_last_error;
void mpe_set_error(const char* what, hipError_t e) { g_last_error = std::string(what) + ": " + hipGetErrorString(e); }
void mpe_set_error_msg(const char* what) { g_last_error = what; }

// =============================================================================================
// core: contexts, workspace, modulus sets, modexp / modmul launches
// =============================================================================================
namespace mpe {

int ws_reserve(mpe_ctx* ctx, size_t bytes, hipStream_t st) {
  if (ctx->ws_hold) {               // several composites share one reservation (they run concurrently): no reset, no move
    if (ctx->ws_off + bytes + ctx->ws_top > ctx->ws_bytes) { mpe_set_error_msg("workspace: the shared reservation is too small"); return MPE_E_NOMEM; }
    return MPE_OK;
  }
  ctx->ws_off = 0;
  if (bytes + ctx->ws_top <= ctx->ws_bytes) return MPE_OK;
  if (ctx->ws_top) { mpe_set_error_msg("workspace: cannot grow while a caller keeps arrays at its top"); return MPE_E_NOMEM; }
  if (ctx->ws) { (void)hipStreamSynchronize(st); (void)hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
  const size_t want = bytes + (bytes >> 2) + (1u << 20);
  hipError_t e = hipMalloc(&ctx->ws, want);
  if (e != hipSuccess) { mpe_set_error("hipMalloc(workspace)", e); return MPE_E_NOMEM; }
  ctx->ws_bytes = want;
  return MPE_OK;
}
void* ws_alloc(mpe_ctx* ctx, size_t bytes) {
  const size_t off = (ctx->ws_off + 255) & ~(size_t)255;
  if (off + bytes + ctx->ws_top > ctx->ws_bytes) return nullptr;     // ws_reserve under-estimated: a library bug
  ctx->ws_off = off + bytes;
  return (char*)ctx->ws + off;
}

static int slot_of(const mpe_ctx* ctx, hipStream_t st) {
  for (int i = 0; i < 3; ++i) if (ctx->aux[i] && st == ctx->aux[i]) return i + 1;
  return 0;
}
uint32_t* tables_for(mpe_ctx* ctx, size_t need, hipStream_t st) {
  const int sl = slot_of(ctx, st);
  if (need > ctx->tables_bytes[sl]) {
    if (ctx->tables[sl]) { (void)hipStreamSynchronize(st); (void)hipFree(ctx->tables[sl]); ctx->tables[sl] = nullptr; ctx->tables_bytes[sl] = 0; }
    hipError_t e = hipMalloc(&ctx->tables[sl], need);
    if (e != hipSuccess) { mpe_set_error("hipMalloc(window tables)", e); return nullptr; }
    ctx->tables_bytes[sl] = need;
  }
  return (uint32_t*)ctx->tables[sl];
}
bool ensure_aux(mpe_ctx* ctx) {
  if (ctx->aux_ready) return true;
  for (int i = 0; i < 3; ++i) if (hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking) != hipSuccess) return false;
  for (int i = 0; i < 2; ++i) if (hipEventCreateWithFlags(&ctx->ev_fork[i], hipEventDisableTiming) != hipSuccess) return false;
  for (int i = 0; i < 3; ++i) if (hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) != hipSuccess) return false;
  if (hipEventCreateWithFlags(&ctx->ev_mid, hipEventDisableTiming) != hipSuccess) return false;
  if (hipEventCreateWithFlags(&ctx->ev_ahead, hipEventDisableTiming) != hipSuccess) return false;
  ctx->aux_ready = true;
  return true;
}
Fork::Fork(mpe_ctx* c, hipStream_t m, int branches, bool enable, int first_aux) : ctx(c), main(m), n(branches), first(first_aux) {
  on = enable && c->allow_par && branches > 1 && first_aux + branches - 1 <= 3 && ensure_aux(c);
  if (!on) return;
  hipEvent_t ev = c->ev_fork[first_aux ? 1 : 0];
  (void)hipEventRecord(ev, m);
  for (int i = 1; i < n; ++i) (void)hipStreamWaitEvent(c->aux[first + i - 1], ev, 0);
}
void Fork::branch_done_wait(int i, hipStream_t waiter) {
  if (!on || i <= 0 || waiter == ctx->aux[first + i - 1]) return;
  (void)hipEventRecord(ctx->ev_join[first + i - 1], ctx->aux[first + i - 1]);
  (void)hipStreamWaitEvent(waiter, ctx->ev_join[first + i - 1], 0);
}
void Fork::join() {
  if (!on) return;
  for (int i = 1; i < n; ++i) {
    (void)hipEventRecord(ctx->ev_join[first + i - 1], ctx->aux[first + i - 1]);
    (void)hipStreamWaitEvent(main, ctx->ev_join[first + i - 1], 0);
  }
}

void prof_begin(mpe_ctx* ctx, hipStream_t st, int kind, int bits, int exp_words, int batch, int exp2_words) {
  if (!ctx->prof_on) return;
  mpe_ctx::ProfEvt ev;
  ev.kind = kind; ev.bits = bits; ev.exp_words = exp_words; ev.batch = batch; ev.exp2_words = exp2_words; ev.wave_trips = 0;
  (void)hipEventCreate(&ev.a);
  (void)hipEventCreate(&ev.b);
  (void)hipEventRecord(ev.a, st);
  ctx->prof.push_back(ev);
}
uint32_t* prof_counter(mpe_ctx* ctx, int wave_trips) {
  if (!ctx->prof_on || ctx->prof.empty() || !ctx->prof_ctr || (int)ctx->prof.size() > ctx->prof_ctr_cap) return nullptr;
  ctx->prof.back().wave_trips = wave_trips;
  return ctx->prof_ctr + (ctx->prof.size() - 1);
}
void prof_end(mpe_ctx* ctx, hipStream_t st) {
  if (ctx->prof_on && !ctx->prof.empty()) (void)hipEventRecord(ctx->prof.back().b, st);
}

static ModsetView view_of(const mpe_modset* ms) {
  ModsetView v;
  v.n_limbs = ms->n_limbs; v.one_limbs = ms->one_limbs; v.r2_limbs = ms->r2_limbs; v.r2h_limbs = ms->r2h_limbs;
  v.n0inv = ms->n0inv; v.count = ms->count;
  return v;
}

// persistent grid: at most the resident-wave capacity; when the batch needs several trips the
// grid shrinks to the smallest one that still finishes in that many trips (no half-empty tail)
template <class C>
static int grid_for(const mpe_ctx* ctx, int batch, int waves_per_cu) {
  return persistent_grid(ctx, (batch + C::GROUPS - 1) / C::GROUPS, ctx->cus * waves_per_cu);
}

template <class C>
static int modset_create_impl(int count, const uint32_t* d_moduli, mpe_modset** out, hipStream_t st) {
  mpe_modset* ms = new (std::nothrow) mpe_modset();
  if (!ms) return MPE_E_NOMEM;
  ms->bits = C::BITS;
  ms->count = count;
  ms->K = C::K;
  const size_t words = (size_t)count * C::K;
  const size_t iw = (size_t)count * C::K32;
  const size_t total = (4 * words + (size_t)count + iw + 4) * sizeof(uint32_t);
  hipError_t e = hipMalloc(&ms->blob, total);
  if (e != hipSuccess) { delete ms; mpe_set_error("hipMalloc(modset)", e); return MPE_E_NOMEM; }
  uint32_t* p = (uint32_t*)ms->blob;
  ms->n_limbs = p;
  ms->one_limbs = p + words;
  ms->r2_limbs = p + 2 * words;
  ms->r2h_limbs = p + 3 * words;
  ms->n0inv = p + 4 * words;
  ms->words = ms->n0inv + count;
  ms->one_words = ms->words + iw;
  (void)hipMemcpyAsync(ms->words, d_moduli, iw * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
  (void)hipMemsetD32Async((hipDeviceptr_t)ms->one_words, 1, 1, st);
  const int blocks = (count + C::GROUPS - 1) / C::GROUPS;
  hipLaunchKernelGGL(modset_setup_kernel<C>, dim3(blocks), dim3(64), 0, st, count, d_moduli, ms->n_limbs,
                     ms->one_limbs, ms->r2_limbs, ms->r2h_limbs, ms->n0inv);
  e = hipGetLastError();
  if (e != hipSuccess) { (void)hipFree(ms->blob); delete ms; mpe_set_error("modset_setup_kernel", e); return MPE_E_HIP; }
  *out = ms;
  return MPE_OK;
}

int modset_create_dev(mpe_ctx* ctx, int bits, int count, const uint32_t* d_moduli, mpe_modset** out, hipStream_t st) {
  (void)ctx;
  if (bits == 4096) return modset_create_impl<Cfg4096>(count, d_moduli, out, st);
  if (bits == 2048) return modset_create_impl<Cfg2048>(count, d_moduli, out, st);
  return MPE_E_ARG;
}

template <class C>
static int modexp_impl(mpe_ctx* ctx, const mpe_modset* ms, int batch, Rows mod_sel, Rows base_lo, Rows base_hi, Rows exps,
                       int exp_words, Rows base2, Rows exps2, int exp2_words, uint32_t* d_out, hipStream_t st) {
  const int units = (batch + C::GROUPS - 1) / C::GROUPS, cap = ctx->cus * ctx->modexp_waves_per_cu;
  const int grid = ladder_grid(ctx, units, cap);
  // window width: multiplications = E + E/wb + 2^wb; 4 bits up to 256-bit exponents, 5 up to ~1500, 6 beyond
  int wb = exp_words <= 8 ? 4 : (exp_words < 48 ? 5 : 6);
  if (ctx->window_bits) wb = ctx->window_bits;
  const bool dual = base2.p != nullptr;
  if (dual && (base_hi.p != nullptr || exp2_words <= 0 || 32 * exp2_words > ((exp_words * 32 + wb - 1) / wb - 1) * wb)) {
    mpe_set_error_msg("modexp: the second exponent must be shorter than the first");
    return MPE_E_ARG;
  }
  const size_t need = (size_t)grid * C::GROUPS * (((size_t)1 << wb) + (dual ? 16 : 0)) * C::K * sizeof(uint32_t);
  uint32_t* tabs = tables_for(ctx, need + SCHED_WORDS * sizeof(int32_t), st);
  if (!tabs) return MPE_E_NOMEM;
  const SchedArgs sched = ladder_sched(ctx, units, cap, (int32_t*)((char*)tabs + need), st);
  prof_begin(ctx, st, 0, C::BITS, exp_words, batch, dual ? exp2_words : 0);
  hipLaunchKernelGGL(modexp_kernel<C>, dim3(grid), dim3(64), 0, st, batch, view_of(ms), mod_sel, base_lo, base_hi, exps,
                     exp_words, wb, base2, exps2, exp2_words, d_out, tabs, sched);
  prof_end(ctx, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("modexp_kernel", e); return MPE_E_HIP; }
  ctx->last.waves = grid;
  ctx->last.ints_per_wave = C::GROUPS;
  ctx->last.limbs = C::K;
  ctx->last.limb_bits = C::W;
  ctx->last.lds_bytes_per_wave = C::LDS_WORDS * 4;
  ctx->last.table_scratch_bytes = need;
  return MPE_OK;
}

int launch_modexp2(mpe_ctx* ctx, const mpe_modset* ms, int batch, Rows mod_sel, Rows base, Rows exps, int exp_words,
                   Rows base2, Rows exps2, int exp2_words, uint32_t* out, hipStream_t st) {
  if (batch == 0) return MPE_OK;
  if (ms->bits == 4096)
    return modexp_impl<Cfg4096>(ctx, ms, batch, mod_sel, base, no_rows(), exps, exp_words, base2, exps2, exp2_words, out, st);
  if (ms->bits == 2048)
    return modexp_impl<Cfg2048>(ctx, ms, batch, mod_sel, base, no_rows(), exps, exp_words, base2, exps2, exp2_words, out, st);
  return MPE_E_ARG;
}
int launch_modexp(mpe_ctx* ctx, const mpe_modset* ms, int batch, Rows mod_sel, Rows base_lo, Rows base_hi, Rows exps,
                  int exp_words, uint32_t* out, hipStream_t st) {
  if (batch == 0) return MPE_OK;
  if (ms->bits == 4096)
    return modexp_impl<Cfg4096>(ctx, ms, batch, mod_sel, base_lo, base_hi, exps, exp_words, no_rows(), no_rows(), 0, out, st);
  if (ms->bits == 2048)
    return modexp_impl<Cfg2048>(ctx, ms, batch, mod_sel, base_lo, base_hi, exps, exp_words, no_rows(), no_rows(), 0, out, st);
  return MPE_E_ARG;
}

template <class C>
static int modmul_impl(mpe_ctx* ctx, const mpe_modset* ms, int batch, Rows mod_sel, Rows a, Rows b, uint32_t* d_out,
                       hipStream_t st) {
  const int grid = grid_for<C>(ctx, batch, 16);
  prof_begin(ctx, st, 1, C::BITS, 0, batch);
  hipLaunchKernelGGL(modmul_kernel<C>, dim3(grid), dim3(64), 0, st, batch, view_of(ms), mod_sel, a, b, d_out);
  prof_end(ctx, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("modmul_kernel", e); return MPE_E_HIP; }
  return MPE_OK;
}

int launch_modmul(mpe_ctx* ctx, const mpe_modset* ms, int batch, Rows mod_sel, Rows a, Rows b, uint32_t* out,
                  hipStream_t st) {
  if (batch == 0) return MPE_OK;
  if (ms->bits == 4096) return modmul_impl<Cfg4096>(ctx, ms, batch, mod_sel, a, b, out, st);
  if (ms->bits == 2048) return modmul_impl<Cfg2048>(ctx, ms, batch, mod_sel, a, b, out, st);
  return MPE_E_ARG;
}

// mod_idx == NULL means: one modulus for everybody when count == 1, else modulus i for item i
static Rows mod_selector(const mpe_modset* ms, const int32_t* d_mod_idx) {
  return Rows{nullptr, d_mod_idx, (d_mod_idx == nullptr && ms->count != 1) ? 1 : 0, 0};
}

}  // namespace mpe

#include "mpe_small.h"

namespace mpe {
// N-adic pair engine: dispatch on the width of the modulus (the kernels live in their own translation units)
static void pairset_free(mpe_pairset* ps) {
  if (!ps) return;
  if (ps->blob) (void)hipFree(ps->blob);
  delete ps;
}
static int pairset_create(int half_bits, int count, const uint32_t* d_moduli, mpe_pairset** out, hipStream_t st) {
  if (half_bits == 2048) return pairset_create_2048(count, d_moduli, out, st);
  if (half_bits == 1024) return pairset_create_1024(count, d_moduli, out, st);
  return MPE_E_ARG;
}
// base^exps [* base2^exps2] modulo the SQUARE of modulus mod_sel(i) of `ps`; out rows are 2 * half_bits/32 words
// half != 0: plain exponentiation modulo the modulus itself (out rows keep the 2 * half_bits/32 layout, value < N)
static int launch_pair_modexp(mpe_ctx* ctx, const mpe_pairset* ps, int batch, Rows mod_sel, Rows base, Rows exps, int exp_words,
                              Rows base2, Rows exps2, int exp2_words, uint32_t* out, hipStream_t st, int half = 0, int public_exp = 0) {
  // public_exp: the caller vouches that `exps` is public (the key N itself): the kernel may then let the operation sequence
  // depend on it (sliding windows).  Everything else (nonces, shares, p, q, challenges of a prover) keeps the fixed schedule.
  if (batch == 0) return MPE_OK;
  if (ps->half_bits == 2048)
    return pair_modexp_2048(ctx, ps, batch, mod_sel, base, exps, exp_words, base2, exps2, exp2_words, half, out, st, public_exp);
  if (ps->half_bits == 1024)
    return pair_modexp_1024(ctx, ps, batch, mod_sel, base, exps, exp_words, base2, exps2, exp2_words, half, out, st, public_exp);
  return MPE_E_ARG;
}
}  // namespace mpe
#include "mpe_paillier.h"
#include "mpe_proofs.h"
#include "mpe_mta.h"
#include "mpe_bob.h"
#include "mpe_gg20.h"
#include "mpe_sample.h"
#include "mpe_pipeline.h"
#include "mpe_comm.h"
#include "mpe_sigma.h"
#include "mpe_blame.h"
#include "mpe_keygen.h"
#include "mpe_lindell.h"

extern "C" {

// 0.4: mpe_prof_rec grew (sliding_frac); 0.5: sampler, pipeline, comm, keygen verdicts, session abort; 0.6: options instead of
// environment switches, per-ticket pass status + grouping rules of the pipeline, status 91, RCCL bound at run time, mpe_comm_library
const char* mpe_version(void) { return "mpecdsa-hip 0.6.0 (gfx950)"; }
const char* mpe_last_error(void) { return g_last_error.c_str(); }

void mpe_encoding_default(mpe_encoding* e) {
  if (!e) return;
  memset(e, 0, sizeof *e);
  e->ck_salt = 0x4B5A656Eu;                         // b"KZen"
  for (int i = 0; i < 4; ++i) e->ord_dlog[i] = e->ord_cdlog[i] = (uint8_t)i;
  for (int i = 0; i < 8; ++i) e->ord_pedersen[i] = e->ord_heg[i] = e->ord_ecddh[i] = (uint8_t)i;
}
static bool is_perm(const uint8_t* o, int n) {
  unsigned seen = 0;
  for (int i = 0; i < n; ++i) { if (o[i] >= n || (seen >> o[i]) & 1u) return false; seen |= 1u << o[i]; }
  return true;
}
int mpe_ctx_set_encoding(mpe_ctx* ctx, const mpe_encoding* e) {
  if (!ctx || !e) return MPE_E_ARG;
  if (e->chain_point > 1 || e->zero_bytes > 1 || e->ck_mask_order > 1 || e->reserved != 0 || !is_perm(e->ord_dlog, 3) ||
      !is_perm(e->ord_pedersen, 5) || !is_perm(e->ord_heg, 7) || !is_perm(e->ord_ecddh, 6) || !is_perm(e->ord_cdlog, 4)) {
    mpe_set_error_msg("mpe_ctx_set_encoding: a flag is not 0/1 or an order is not a permutation");
    return MPE_E_ARG;
  }
  ctx->enc = *e;
  return MPE_OK;
}
int mpe_ctx_set_device_share(mpe_ctx* ctx, int contexts) {
  if (!ctx || contexts < 1 || contexts > 64) return MPE_E_ARG;
  ctx->device_share = contexts;
  return MPE_OK;
}
int mpe_ctx_get_encoding(const mpe_ctx* ctx, mpe_encoding* out) {
  if (!ctx || !out) return MPE_E_ARG;
  *out = ctx->enc;
  return MPE_OK;
}

// ---- run-time options (A/B switches of the measurements; none changes a result) -----------------------------------------------
namespace mpe {
struct CtxOption {
  const char* key;
  long lo, hi;                                   // accepted integer range (lo == hi == 0: an enumerated string)
  void (*set)(mpe_ctx*, long);
  long (*get)(const mpe_ctx*);
};
#define MPE_OPT_BOOL_OFF(name, field) {name, 0, 1, [](mpe_ctx* c, long v) { c->field = !v; }, [](const mpe_ctx* c) -> long { return c->field ? 0 : 1; }}
#define MPE_OPT_INT(name, lo, hi, field) {name, lo, hi, [](mpe_ctx* c, long v) { c->field = (decltype(c->field))v; }, [](const mpe_ctx* c) -> long { return (long)c->field; }}
static const CtxOption kCtxOptions[] = {
    MPE_OPT_BOOL_OFF("no_fixed_base", use_fixed_base),       // no window tables for h1, h2
    MPE_OPT_BOOL_OFF("no_crt", use_crt),                     // the key holder computes like a peer
    MPE_OPT_BOOL_OFF("no_multiexp", use_multiexp),           // no two-base ladders
    MPE_OPT_BOOL_OFF("no_pair", use_pair),                   // 4096-bit Montgomery kernel instead of the N-adic pair engine
    MPE_OPT_BOOL_OFF("no_pown", use_pown),                   // no x^N = a^p shortcut
    MPE_OPT_BOOL_OFF("no_sliding", use_sliding),             // fixed windows for the public exponent too
    MPE_OPT_BOOL_OFF("no_par", allow_par),                   // one stream, no forks
    MPE_OPT_BOOL_OFF("no_wide", adaptive_lanes),             // 18 limbs per lane always
    MPE_OPT_BOOL_OFF("no_ec_lane_groups", ec_lane_groups),   // one item per lane in the EC round kernels
    MPE_OPT_BOOL_OFF("no_merge_xn", merge_xn),               // round 0's x^N in separate launches
    MPE_OPT_BOOL_OFF("no_merge_r1", merge_r1),               // round 1's verification and MessageB ladders in separate launches
    MPE_OPT_INT("fb_window_bits", 4, 16, fb_window_bits),
    MPE_OPT_INT("window_bits", 0, 6, window_bits),           // 0 = per exponent length
    MPE_OPT_INT("wide_div", 1, 64, wide_div),
    MPE_OPT_INT("merge_r1_quarters", 0, 64, merge_r1_quarters),
    MPE_OPT_INT("no_r1_inversion_ahead", 0, 1, no_r1_inversion_ahead),
    MPE_OPT_BOOL_OFF("no_prio", use_prio),                   // no s_setprio anywhere
    MPE_OPT_INT("no_pdl_ahead", 0, 1, no_pdl_ahead),
    MPE_OPT_BOOL_OFF("no_crt_n", use_crt_n),                 // the provers' r^e mod N on the 2048-bit ladder
    MPE_OPT_INT("no_r1_dlog_first", 0, 1, no_r1_dlog_first),
    MPE_OPT_INT("xwide_div", 0, 1 << 20, xwide_div),
    MPE_OPT_INT("waves_per_cu", 1, 8, modexp_waves_per_cu),
    MPE_OPT_INT("grid_mode", 0, 2, grid_mode),               // 0 equal trips, 1 full trips + tail, 2 hybrid ("grid" takes the names)
    MPE_OPT_INT("fb_split", 0, 64, fb_split),                // lane groups per fixed-base item (0 = chosen per launch)
    MPE_OPT_INT("gg20_trace", 0, 1, gg20_trace),             // synchronise and report after every composite of a round (stderr)
    MPE_OPT_INT("sampler_max_attempts", 1, 1 << 20, sampler_max_attempts),
    MPE_OPT_INT("no_primaries", 0, 1, no_primaries),         // lone launches keep static units; multi-pass launches still pull from the queue
    MPE_OPT_INT("no_elect", 0, 1, no_elect),                 // ladder launches take the dispatcher's placement as it comes (mpe_sched.h)
    {"fb_budget_mb", 0, 1 << 20, [](mpe_ctx* c, long v) { c->fb_budget_bytes = (size_t)v << 20; }, [](const mpe_ctx* c) -> long { return (long)(c->fb_budget_bytes >> 20); }},
};
#undef MPE_OPT_BOOL_OFF
#undef MPE_OPT_INT
extern "C++" void ctx_copy_options(mpe_ctx* dst, const mpe_ctx* src) {      // (this block sits inside the file's extern "C")
  for (const CtxOption& o : kCtxOptions) o.set(dst, o.get(src));
}
}  // namespace mpe

int mpe_ctx_set_option(mpe_ctx* ctx, const char* key, const char* value) {
  if (!ctx || !key || !value) return MPE_E_ARG;
  if (!strcmp(key, "grid")) {
    const int m = !strcmp(value, "equal") ? 0 : (!strcmp(value, "full") ? 1 : (!strcmp(value, "hybrid") ? 2 : -1));
    if (m < 0) { mpe_set_error_msg("mpe_ctx_set_option: grid = equal | full | hybrid"); return MPE_E_ARG; }
    ctx->grid_mode = m;
    return MPE_OK;
  }
  if (!strcmp(key, "no_adaptive_lanes")) {                    // = no_wide + no_ec_lane_groups
    const bool off = atoi(value) != 0;
    ctx->adaptive_lanes = !off; ctx->ec_lane_groups = !off;
    return MPE_OK;
  }
  for (const mpe::CtxOption& o : mpe::kCtxOptions) {
    if (strcmp(key, o.key)) continue;
    char* end = nullptr;
    const long v = strtol(value, &end, 10);
    if (end == value || *end || v < o.lo || v > o.hi) {
      mpe_set_error_msg((std::string("mpe_ctx_set_option: ") + key + " takes an integer in [" + std::to_string(o.lo) + ", " + std::to_string(o.hi) + "]").c_str());
      return MPE_E_ARG;
    }
    o.set(ctx, v);
    return MPE_OK;
  }
  mpe_set_error_msg((std::string("mpe_ctx_set_option: unknown option ") + key).c_str());
  return MPE_E_ARG;
}
int mpe_ctx_get_option(const mpe_ctx* ctx, const char* key, long* value) {
  if (!ctx || !key || !value) return MPE_E_ARG;
  for (const mpe::CtxOption& o : mpe::kCtxOptions)
    if (!strcmp(key, o.key)) { *value = o.get(ctx); return MPE_OK; }
  return MPE_E_ARG;
}
int mpe_ctx_option_count(void) { return (int)(sizeof(mpe::kCtxOptions) / sizeof(mpe::kCtxOptions[0])); }
const char* mpe_ctx_option_name(int i) { return (i >= 0 && i < mpe_ctx_option_count()) ? mpe::kCtxOptions[i].key : nullptr; }

int mpe_ctx_create(mpe_ctx** out, int device) {
  if (!out) return MPE_E_ARG;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { mpe_set_error("hipSetDevice", e); return MPE_E_HIP; }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { mpe_set_error("hipGetDeviceProperties", e); return MPE_E_HIP; }
  mpe_ctx* c = new (std::nothrow) mpe_ctx();
  if (!c) return MPE_E_NOMEM;
  c->device = device;
  c->cus = prop.multiProcessorCount;
  mpe_encoding_default(&c->enc);
  // (the library reads NO environment variable: every A/B switch of the measurements is an option set through
  //  mpe_ctx_set_option below — the Python harness maps its MPE_* variables onto it, tests and tools call it directly)
  // comb tables of the two fixed secp256k1 generators: module globals, built ONCE per device (immutable afterwards — the only
  // process-wide state of the library; a second context never rewrites them under the kernels of the first)
  static std::once_flag comb_once[64];
  static hipError_t comb_err[64];
  if (device < 0 || device >= 64) { delete c; return MPE_E_ARG; }
  std::call_once(comb_once[device], [device]() {
    hipLaunchKernelGGL(mpe::ec::ec_comb_build_kernel, dim3(2), dim3(64), 0, 0);
    comb_err[device] = hipDeviceSynchronize();
  });
  e = comb_err[device];
  if (e != hipSuccess) { mpe_set_error("ec_comb_build_kernel", e); delete c; return MPE_E_HIP; }
  *out = c;
  return MPE_OK;
}

// Zeroes every scratch buffer the context owns (window tables, composite workspace): they hold powers of secret bases and
// nonce-derived intermediates of the last calls (the reference zeroizes its round-1 secrets, range_proofs.rs:26-36,197-212).
int mpe_ctx_wipe(mpe_ctx* ctx, void* stream) {
  if (!ctx) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < 6; ++i) if (ctx->tables[i]) (void)hipMemsetAsync(ctx->tables[i], 0, ctx->tables_bytes[i], st);
  if (ctx->ws) (void)hipMemsetAsync(ctx->ws, 0, ctx->ws_bytes, st);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mpe_set_error("mpe_ctx_wipe", e); return MPE_E_HIP; }
  return MPE_OK;
}

// Audit of the wiping: counts the non-zero 32-bit words in every scratch region the context owns (window tables, composite
// workspace, cached session arena, message slabs).  Synchronises the stream.
__global__ void count_nonzero_kernel(const uint32_t* __restrict__ p, size_t words, unsigned long long* __restrict__ out) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  unsigned long long c = 0;
  for (; g < words; g += step) c += p[g] != 0u;
  if (c) atomicAdd(out, c);
}
int mpe_ctx_scratch_audit(mpe_ctx* ctx, uint64_t* nonzero_words, uint64_t* total_bytes, void* stream) {
  if (!ctx || !nonzero_words) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, 8);
  if (e != hipSuccess) { mpe_set_error("hipMalloc(audit)", e); return MPE_E_NOMEM; }
  (void)hipMemsetAsync(d, 0, 8, st);
  const void* ptrs[11] = {ctx->tables[0], ctx->tables[1], ctx->tables[2], ctx->tables[3], ctx->tables[4], ctx->tables[5], ctx->ws,
                          nullptr, nullptr, ctx->sess_buf, ctx->slab_buf};
  const size_t bytes[11] = {ctx->tables_bytes[0], ctx->tables_bytes[1], ctx->tables_bytes[2], ctx->tables_bytes[3], ctx->tables_bytes[4],
                            ctx->tables_bytes[5], ctx->ws_bytes, 0, 0, ctx->sess_bytes, ctx->slab_bytes};
  uint64_t tot = 0;
  for (int i = 0; i < 11; ++i) {
    if (!ptrs[i] || !bytes[i]) continue;
    tot += bytes[i];
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(4096), dim3(256), 0, st, (const uint32_t*)ptrs[i], bytes[i] / 4, d);
  }
  unsigned long long h = 0;
  e = hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  if (e != hipSuccess) { mpe_set_error("mpe_ctx_scratch_audit", e); return MPE_E_HIP; }
  *nonzero_words = h;
  if (total_bytes) *total_bytes = tot;
  return MPE_OK;
}

int mpe_ctx_destroy(mpe_ctx* ctx) {
  if (!ctx) return MPE_E_ARG;
  (void)mpe_ctx_wipe(ctx, nullptr);
  if (ctx->sess_buf) { (void)hipMemsetAsync(ctx->sess_buf, 0, ctx->sess_bytes, nullptr); }
  if (ctx->slab_buf) { (void)hipMemsetAsync(ctx->slab_buf, 0, ctx->slab_bytes, nullptr); }
  (void)hipDeviceSynchronize();
  if (ctx->sess_buf) (void)hipFree(ctx->sess_buf);
  if (ctx->slab_buf) (void)hipFree(ctx->slab_buf);
  for (int i = 0; i < 6; ++i) if (ctx->tables[i]) (void)hipFree(ctx->tables[i]);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->aux_ready) {
    for (int i = 0; i < 3; ++i) { (void)hipStreamDestroy(ctx->aux[i]); (void)hipEventDestroy(ctx->ev_join[i]); }
    for (int i = 0; i < 2; ++i) (void)hipEventDestroy(ctx->ev_fork[i]);
    if (ctx->ev_mid) (void)hipEventDestroy(ctx->ev_mid);
    if (ctx->ev_ahead) (void)hipEventDestroy(ctx->ev_ahead);
  }
  for (auto& ev : ctx->prof) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  if (ctx->prof_ctr) (void)hipFree(ctx->prof_ctr);
  delete ctx;
  return MPE_OK;
}

int mpe_sync(mpe_ctx* ctx, void* stream) {
  (void)ctx;
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) { mpe_set_error("hipStreamSynchronize", e); return MPE_E_HIP; }
  return MPE_OK;
}

int mpe_last_launch_info(const mpe_ctx* ctx, mpe_launch_info* out) {
  if (!ctx || !out) return MPE_E_ARG;
  *out = ctx->last;
  return MPE_OK;
}

int mpe_prof_enable(mpe_ctx* ctx, int on) {
  if (!ctx) return MPE_E_ARG;
  for (auto& ev : ctx->prof) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  ctx->prof.clear();
  ctx->prof_on = on != 0;
  if (ctx->prof_on) {
    if (!ctx->prof_ctr) {
      ctx->prof_ctr_cap = 1 << 16;
      if (hipMalloc((void**)&ctx->prof_ctr, (size_t)ctx->prof_ctr_cap * sizeof(uint32_t)) != hipSuccess) { ctx->prof_ctr = nullptr; ctx->prof_ctr_cap = 0; }
    }
    if (ctx->prof_ctr) (void)hipMemset(ctx->prof_ctr, 0, (size_t)ctx->prof_ctr_cap * sizeof(uint32_t));
  }
  return MPE_OK;
}

int mpe_prof_collect(mpe_ctx* ctx, mpe_prof_rec* out, int max_records, int* n_out) {
  if (!ctx || !out || !n_out) return MPE_E_ARG;
  int n = 0;
  std::vector<uint32_t> ctr;
  size_t ix = 0;
  // the audit kernels that fill the counters are queued AFTER each record's end event, possibly on non-blocking streams: the
  // counters are final only when the device is idle (profiling runs only — never on the signing path)
  if (ctx->prof_ctr && !ctx->prof.empty()) (void)hipDeviceSynchronize();
  for (auto& ev : ctx->prof) {
    const size_t my = ix++;
    if (n >= max_records) break;
    if (hipEventSynchronize(ev.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev.a, ev.b) != hipSuccess) continue;
    if (ctr.empty() && ctx->prof_ctr) {               // the device was synchronised above: the counters are final
      ctr.resize(ctx->prof.size() < (size_t)ctx->prof_ctr_cap ? ctx->prof.size() : (size_t)ctx->prof_ctr_cap);
      if (hipMemcpy(ctr.data(), ctx->prof_ctr, ctr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) ctr.assign(ctr.size(), 0u);
    }
    out[n].kind = ev.kind; out[n].bits = ev.bits; out[n].exp_words = ev.exp_words; out[n].batch = ev.batch; out[n].ms = ms;
    out[n].exp2_words = ev.exp2_words;
    out[n].sliding_frac = (ev.wave_trips > 0 && my < ctr.size()) ? (float)ctr[my] / (float)ev.wave_trips : (ev.kind == 6 ? -1.f : 0.f);
    ++n;
  }
  *n_out = n;
  return MPE_OK;
}

int mpe_modset_count(const mpe_modset* ms) { return ms ? ms->count : MPE_E_ARG; }
int mpe_modset_bits(const mpe_modset* ms) { return ms ? ms->bits : MPE_E_ARG; }

int mpe_modset_destroy(mpe_modset* ms) {
  if (!ms) return MPE_E_ARG;
  if (ms->blob) (void)hipFree(ms->blob);
  delete ms;
  return MPE_OK;
}

int mpe_modset_create(mpe_ctx* ctx, int bits, int count, const uint32_t* d_moduli, mpe_modset** out, void* stream) {
  if (!ctx || !out || !d_moduli || count <= 0) return MPE_E_ARG;
  return modset_create_dev(ctx, bits, count, d_moduli, out, (hipStream_t)stream);
}

int mpe_modexp(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx, const uint32_t* d_base,
               const uint32_t* d_exp, int exp_words, uint32_t* d_out, void* stream) {
  if (!ctx || !ms || !d_base || !d_exp || !d_out || batch < 0 || exp_words <= 0) return MPE_E_ARG;
  if (!d_mod_idx && ms->count != 1 && ms->count < batch) return MPE_E_ARG;
  const int k32 = ms->bits / 32;
  return launch_modexp(ctx, ms, batch, mod_selector(ms, d_mod_idx), rows(d_base, k32), no_rows(), rows(d_exp, exp_words),
                       exp_words, d_out, (hipStream_t)stream);
}

int mpe_modexp2(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx, const uint32_t* d_base,
                const uint32_t* d_exp, int exp_words, const uint32_t* d_base2, const uint32_t* d_exp2, int exp2_words,
                uint32_t* d_out, void* stream) {
  if (!ctx || !ms || !d_base || !d_exp || !d_base2 || !d_exp2 || !d_out || batch < 0 || exp_words <= 0 || exp2_words <= 0)
    return MPE_E_ARG;
  if (!d_mod_idx && ms->count != 1 && ms->count < batch) return MPE_E_ARG;
  const int k32 = ms->bits / 32;
  return launch_modexp2(ctx, ms, batch, mod_selector(ms, d_mod_idx), rows(d_base, k32), rows(d_exp, exp_words), exp_words,
                        rows(d_base2, k32), rows(d_exp2, exp2_words), exp2_words, d_out, (hipStream_t)stream);
}

int mpe_modmul(mpe_ctx* ctx, const mpe_modset* ms, int batch, const int32_t* d_mod_idx, const uint32_t* d_a,
               const uint32_t* d_b, uint32_t* d_out, void* stream) {
  if (!ctx || !ms || !d_a || !d_b || !d_out || batch < 0) return MPE_E_ARG;
  if (!d_mod_idx && ms->count != 1 && ms->count < batch) return MPE_E_ARG;
  const int k32 = ms->bits / 32;
  return launch_modmul(ctx, ms, batch, mod_selector(ms, d_mod_idx), rows(d_a, k32), rows(d_b, k32), d_out,
                       (hipStream_t)stream);
}

}  // extern "C"
