// A signing SERVICE's view of the lock-step composition: batches of `batch` sessions arrive one after another (BASELINE config 4's
// literal shape is a stream of 1 024-session batches) and must not be latency-bound one by one.  The reference runs many
// `OfflineStage`s concurrently on one executor (round_based's AsyncProtocol; state_machine/sign.rs:667-691 runs its parties through
// `Simulation`, and rounds.rs:106,215,323 mark the rounds `is_expensive` so that they go to a blocking pool); the device equivalent:
//
//   * `group` consecutive batches are coalesced into ONE pass of mpe_gg20_sign (a "super-batch"): every heavy launch carries the
//     items of all of them, so the launches are large enough for the efficient lane layouts and fill whole passes of the chip
//     (the merged round-1 ladder launch of 4 x 1 024 sessions is 49 152 + 16 384 items = two full passes of 32 768 resident groups);
//   * `lanes` super-batches are in flight at once, each on ONE stream of its own (no forked streams inside a lane): the
//     latency-bound stretches of one (EC kernels, short ladders, inversions) run under the throughput-bound ladders of another.
//     lanes <= 4 streams in total, so the runtime's default 4 hardware queues never multiplex two lanes onto one queue —
//     no GPU_MAX_HW_QUEUES, no child process (round 4 needed 16 queues and 8 host threads for 13.5 k signatures/s;
//     profiles/r05/pipeline_sweep.jsonl has this design's numbers);
//   * ONE host thread feeds everything: submit() only enqueues — the caller's arrays are copied into the lane's staging arrays
//     behind the lane's running pass, or (submit_seeded) nothing is copied at all: the values of ALL batches of a group are drawn
//     by ONE launch of the device-side sampler (mpe_sample.h) right before the pass.  (Staging the inputs on a stream of their own
//     was built and measured: the sampler's small kernels starve behind the other lane's persistent ladders and the lanes wait for
//     them — 14.8 k against 15.8 k signatures/s; dropped.)
// Results are bit-identical to mpe_gg20_sign on the same inputs: a session's outputs do not depend on its neighbours in a batch.
//
// Failure contract (round 6; the reference reports per session and never loses an error: gg_2020/mod.rs:23-27, rounds.rs:696-713):
// when the pass of a group fails as a whole (MPE_E_NOMEM / MPE_E_HIP from the sampler or from mpe_gg20_sign) EVERY batch of the
// group learns it — its d_status is filled with MPE_GG20_STATUS_PASS_FAILED(rc) and its r / s / recid (/ R) zeroed on the lane's
// stream, the ticket keeps rc, and wait / query / stream_wait / latency_ms / pass_ms of that ticket return it.  The next group is
// not affected.  mpe_gg20_pipeline_inject_fault makes the next passes fail that way (tests).
//
// When does a part-filled group go?  (1) full; (2) flush / wait on one of its tickets; (3) `deadline`: its oldest batch has
// waited T microseconds of host time; (4) `eager`: the lane it would run on is idle — arrival-driven grouping: under a trickle
// every batch starts at once (lowest latency), under load the lanes are busy and the groups fill by themselves.  (3) and (4)
// are evaluated inside every submit / query / poll call — the ONE host thread that drives the pipeline; there is no hidden thread.
// A new group opens on an idle lane if there is one, otherwise on the next lane in turn.
// Included by mpe_lib.hip after mpe_sample.h.
#pragma once
#include <chrono>

#include "mpe_sample.h"

struct mpe_gg20_pipeline {
  static constexpr int MAX_LANES = 4, MAX_GROUP = 64, RING = 4096;
  mpe_ctx* parent = nullptr;
  const mpe_gg20_keys* K = nullptr;
  int batch = 0, group = 0, lanes = 0, dedup = 0;
  int64_t deadline_us = -1;                   // < 0: no deadline
  int eager = 0;                              // launch a part-filled group as soon as its lane is idle
  int fail_next = 0, fail_rc = MPE_E_NOMEM;   // fault injection: that many next passes fail with fail_rc
  using Clock = std::chrono::steady_clock;
  struct Out { uint32_t *r, *s, *R; int32_t *recid, *status; };
  struct Lane {
    mpe_ctx* ctx = nullptr;
    hipStream_t st = nullptr;
    mpe_gg20_nonce_buf* stage = nullptr;      // [group * batch] sessions, every signer local
    int32_t* keyset = nullptr;
    void* blob = nullptr;                     // keyset | r | s | R | recid | status | fail staging
    size_t blob_bytes = 0;
    Out res{};
    int32_t* fail = nullptr;
    int filled = 0, seeded = 0;               // seeded: slots of the open group whose values the sampler draws (all or none)
    bool any_keyset = false;
    uint8_t seeds[32 * mpe::smp::MAX_SLOTS];
    uint64_t counters[mpe::smp::MAX_SLOTS];
    Out dst[MAX_GROUP];
    uint64_t ticket[MAX_GROUP];
    uint64_t last_ticket = 0;                 // the last batch of the last pass queued on this lane (idle test)
    Clock::time_point oldest;                 // host time of the open group's first submission
  } lane[MAX_LANES];
  struct Ticket { uint64_t id = 0, begin_of = 0; hipEvent_t submitted = nullptr, begin = nullptr, done = nullptr; int lane = -1, rc = MPE_OK; bool launched = false, used = false; };
  Ticket ring[RING];
  int cur = 0;
  uint64_t next_ticket = 1;
  uint64_t launched_groups = 0, launched_by_deadline = 0, launched_by_idle = 0, failed_groups = 0;
};

namespace mpe {
namespace pipe {

static mpe_gg20_pipeline::Ticket* ticket_of(mpe_gg20_pipeline* p, uint64_t id) {
  if (id == 0 || id >= p->next_ticket) return nullptr;
  mpe_gg20_pipeline::Ticket* t = &p->ring[id % mpe_gg20_pipeline::RING];
  return (t->used && t->id == id) ? t : nullptr;
}

// every ticket of a failed pass reports it: status = MPE_GG20_STATUS_PASS_FAILED(rc) for every session, no signature
__global__ void fill_i32_kernel(int32_t* __restrict__ p, int n, int32_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
static void wipe_bytes(void* p, size_t n) {                   // not elidable
  volatile uint8_t* v = (volatile uint8_t*)p;
  for (size_t i = 0; i < n; ++i) v[i] = 0;
}
static bool lane_idle(mpe_gg20_pipeline* p, int li) {
  mpe_gg20_pipeline::Lane& L = p->lane[li];
  if (!L.last_ticket) return true;
  mpe_gg20_pipeline::Ticket* t = ticket_of(p, L.last_ticket);
  return !t || !t->launched || hipEventQuery(t->done) == hipSuccess;      // (a recycled ring slot: that pass ended long ago)
}

// the open group of lane `li` goes to the device: one pass over filled * batch sessions, then every batch's results to its owner
static int launch_group(mpe_gg20_pipeline* p, int li) {
  mpe_gg20_pipeline::Lane& L = p->lane[li];
  if (L.filled == 0) return MPE_OK;
  const int B = L.filled * p->batch;
  mpe_gg20_nonces Z;
  (void)mpe_gg20_nonces_view(L.stage, &Z);
  const int32_t* ks = (p->K->K > 1 || L.any_keyset) ? L.keyset : nullptr;
  if (mpe_gg20_pipeline::Ticket* t0 = ticket_of(p, L.ticket[0])) (void)hipEventRecord(t0->begin, L.st);
  int rc = MPE_OK;
  if (L.seeded) {                               // every value of every batch of the group: one launch of the sampler
    int32_t local[8];
    for (int i = 0; i < p->K->S; ++i) local[i] = i;
    rc = mpe::smp::sample_gg20(L.ctx, p->K, p->batch, L.filled, p->K->S, local, ks, L.seeds, L.counters, &Z, L.fail, L.st);
    // the seeds derive every nonce of their batches: as sensitive as the key share, gone from host memory once the sampler is queued
    // (the kernel arguments were copied at launch)
    wipe_bytes(L.seeds, sizeof L.seeds);
    wipe_bytes(L.counters, sizeof L.counters);
  }
  if (rc == MPE_OK && p->fail_next > 0) {       // fault injection (tests): the pass fails as a whole, after its inputs were staged
    --p->fail_next;
    rc = p->fail_rc;
    mpe_set_error_msg("gg20 pipeline: injected pass failure");
  }
  if (rc == MPE_OK)
    rc = mpe_gg20_sign(L.ctx, p->K, B, ks, &Z, L.res.r, L.res.s, L.res.recid, L.res.R, L.res.status, p->dedup, 0, L.st);
  if (rc == MPE_OK) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mpe_set_error("gg20 pipeline launch", e); rc = MPE_E_HIP; }
  } else {
    (void)hipGetLastError();
  }
  for (int g = 0; g < L.filled; ++g) {
    const size_t o = (size_t)g * p->batch, nb = (size_t)p->batch;
    const mpe_gg20_pipeline::Out& d = L.dst[g];
    if (rc == MPE_OK) {
      (void)hipMemcpyAsync(d.r, L.res.r + o * 8, nb * 32, hipMemcpyDeviceToDevice, L.st);
      (void)hipMemcpyAsync(d.s, L.res.s + o * 8, nb * 32, hipMemcpyDeviceToDevice, L.st);
      (void)hipMemcpyAsync(d.recid, L.res.recid + o, nb * 4, hipMemcpyDeviceToDevice, L.st);
      (void)hipMemcpyAsync(d.status, L.res.status + o, nb * 4, hipMemcpyDeviceToDevice, L.st);
      if (d.R) (void)hipMemcpyAsync(d.R, L.res.R + o * 16, nb * 64, hipMemcpyDeviceToDevice, L.st);
    } else {                                    // the owner of EVERY batch of the group reads the failure in its own arrays
      hipLaunchKernelGGL(fill_i32_kernel, dim3(mpe::blocks_for((int)nb, 256)), dim3(256), 0, L.st, d.status, (int)nb, (int32_t)MPE_GG20_STATUS_PASS_FAILED(rc));
      (void)hipMemsetAsync(d.r, 0, nb * 32, L.st);
      (void)hipMemsetAsync(d.s, 0, nb * 32, L.st);
      (void)hipMemsetAsync(d.recid, 0, nb * 4, L.st);
      if (d.R) (void)hipMemsetAsync(d.R, 0, nb * 64, L.st);
    }
  }
  // the staged sampled values (k_i, gamma_i, Paillier randomness) of this group do not wait for the next one to overwrite them,
  // nor do the lane's copies of the signatures (the owners have theirs)
  (void)hipMemsetAsync(L.stage->blob, 0, L.stage->bytes, L.st);
  (void)hipMemsetAsync(L.res.r, 0, (size_t)B * 8 * 4, L.st);
  (void)hipMemsetAsync(L.res.s, 0, (size_t)B * 8 * 4, L.st);
  mpe_gg20_pipeline::Ticket* t0 = ticket_of(p, L.ticket[0]);
  for (int g = 0; g < L.filled; ++g) {
    mpe_gg20_pipeline::Ticket* t = ticket_of(p, L.ticket[g]);
    if (!t) continue;
    (void)hipEventRecord(t->done, L.st);
    if (g && t0) t->begin_of = t0->id;
    t->rc = rc;
    t->launched = true;
  }
  L.last_ticket = L.ticket[L.filled - 1];
  L.filled = 0;
  L.seeded = 0;
  L.any_keyset = false;
  p->launched_groups++;
  if (rc != MPE_OK) p->failed_groups++;
  // the next group opens on an idle lane if there is one, otherwise on the next lane in turn
  int next = (li + 1) % p->lanes;
  for (int k = 0; k < p->lanes; ++k) {
    const int c = (li + 1 + k) % p->lanes;
    if (lane_idle(p, c)) { next = c; break; }
  }
  p->cur = next;
  return rc;
}

// rules (3) and (4) of the header comment for the open group; *launched += 1 when it went
static int poll_open_group(mpe_gg20_pipeline* p, int* launched) {
  mpe_gg20_pipeline::Lane& L = p->lane[p->cur];
  if (L.filled == 0) return MPE_OK;
  bool go = false;
  if (p->eager && lane_idle(p, p->cur)) { go = true; p->launched_by_idle++; }
  if (!go && p->deadline_us >= 0 &&
      std::chrono::duration_cast<std::chrono::microseconds>(mpe_gg20_pipeline::Clock::now() - L.oldest).count() >= p->deadline_us) { go = true; p->launched_by_deadline++; }
  if (!go) return MPE_OK;
  if (launched) ++*launched;
  return launch_group(p, p->cur);
}

// common part of the two submit forms: reserves slot g of the current lane, records the ticket
static int open_slot(mpe_gg20_pipeline* p, uint32_t* d_r, uint32_t* d_s, int32_t* d_recid, uint32_t* d_R, int32_t* d_status, hipStream_t caller,
                     int* lane_out, int* slot_out, uint64_t* ticket_out) {
  if (!p || !d_r || !d_s || !d_recid || !d_status || !ticket_out) return MPE_E_ARG;
  const uint64_t id = p->next_ticket;
  mpe_gg20_pipeline::Ticket* t = &p->ring[id % mpe_gg20_pipeline::RING];
  if (t->used && (!t->launched || hipEventQuery(t->done) != hipSuccess)) {
    mpe_set_error_msg("gg20 pipeline: more than 4096 batches in flight");
    return MPE_E_ARG;
  }
  for (hipEvent_t* ev : {&t->submitted, &t->begin, &t->done})        // created once per ring slot (a failed creation is retried by the next use)
    if (!*ev && hipEventCreate(ev) != hipSuccess) { *ev = nullptr; mpe_set_error_msg("gg20 pipeline: hipEventCreate"); return MPE_E_HIP; }
  const int li = p->cur;
  mpe_gg20_pipeline::Lane& L = p->lane[li];
  const int g = L.filled;
  t->id = id; t->begin_of = id; t->lane = li; t->rc = MPE_OK; t->launched = false; t->used = true;
  if (g == 0) L.oldest = mpe_gg20_pipeline::Clock::now();
  // the lane reads the caller's arrays only after the caller's stream has produced them
  (void)hipEventRecord(t->submitted, caller);
  (void)hipStreamWaitEvent(L.st, t->submitted, 0);
  L.dst[g] = mpe_gg20_pipeline::Out{d_r, d_s, d_R, d_recid, d_status};
  L.ticket[g] = id;
  p->next_ticket++;
  *lane_out = li; *slot_out = g; *ticket_out = id;
  return MPE_OK;
}
static int close_slot(mpe_gg20_pipeline* p, int li, const int32_t* d_keyset) {
  mpe_gg20_pipeline::Lane& L = p->lane[li];
  const size_t o = (size_t)L.filled * p->batch;
  if (d_keyset) { (void)hipMemcpyAsync(L.keyset + o, d_keyset, (size_t)p->batch * 4, hipMemcpyDeviceToDevice, L.st); L.any_keyset = true; }
  else (void)hipMemsetAsync(L.keyset + o, 0, (size_t)p->batch * 4, L.st);
  L.filled++;
  // a pass that fails is reported through the tickets of its batches (each owner learns it from wait / query and from its
  // status array), not through whichever submit happened to close the group: the submission itself succeeded
  if (L.filled == p->group) (void)launch_group(p, li);
  else (void)poll_open_group(p, nullptr);
  return MPE_OK;
}

}  // namespace pipe
}  // namespace mpe

extern "C" {

int mpe_gg20_pipeline_destroy(mpe_gg20_pipeline* p);

int mpe_gg20_pipeline_create(mpe_ctx* ctx, const mpe_gg20_keys* keys, int batch, int group, int lanes, int dedup_verify, mpe_gg20_pipeline** out) {
  if (!ctx || !keys || !out || batch <= 0 || group < 1 || group > mpe_gg20_pipeline::MAX_GROUP || lanes < 1 || lanes > mpe_gg20_pipeline::MAX_LANES)
    return MPE_E_ARG;
  for (int i = 0; i < keys->S; ++i) if (keys->own_slot[keys->signers[i]] < 0) { mpe_set_error_msg("gg20 pipeline: the key object must hold every signer's secrets"); return MPE_E_ARG; }
  mpe_gg20_pipeline* p = new (std::nothrow) mpe_gg20_pipeline();
  if (!p) return MPE_E_NOMEM;
  p->parent = ctx; p->K = keys; p->batch = batch; p->group = group; p->lanes = lanes; p->dedup = dedup_verify ? 1 : 0;
  const size_t GB = (size_t)group * batch;
  int rc = MPE_OK;
  for (int li = 0; li < lanes && rc == MPE_OK; ++li) {
    mpe_gg20_pipeline::Lane& L = p->lane[li];
    rc = mpe_ctx_create(&L.ctx, ctx->device);
    if (rc != MPE_OK) break;
    // a lane is the parent context with a workspace of its own and NO forked streams (one stream per lane: <= 4 in total)
    mpe_ctx* c = L.ctx;
    mpe::ctx_copy_options(c, ctx);            // every option of the parent (mpe_ctx_set_option), then:
    c->enc = ctx->enc;
    c->allow_par = false;
    c->use_prio = 0;                          // the lanes' ladders compete with each other, not with side work of their own pass
    if (hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking) != hipSuccess) { mpe_set_error_msg("gg20 pipeline: hipStreamCreate"); rc = MPE_E_HIP; break; }
    rc = mpe_gg20_nonces_alloc(c, keys, (int)GB, keys->S, &L.stage);
    if (rc != MPE_OK) break;
    const size_t words = GB * (1 + 8 + 8 + 16 + 1 + 1) + 64;
    L.blob_bytes = words * 4;
    if (hipMalloc(&L.blob, L.blob_bytes) != hipSuccess) { mpe_set_error_msg("gg20 pipeline: hipMalloc(staging)"); rc = MPE_E_NOMEM; break; }
    (void)hipMemset(L.blob, 0, L.blob_bytes);
    uint32_t* w = (uint32_t*)L.blob;
    L.keyset = (int32_t*)w; w += GB;
    L.res.r = w; w += GB * 8; L.res.s = w; w += GB * 8; L.res.R = w; w += GB * 16;
    L.res.recid = (int32_t*)w; w += GB; L.res.status = (int32_t*)w; w += GB;
    L.fail = (int32_t*)w;
  }
  if (rc != MPE_OK) { mpe_gg20_pipeline_destroy(p); return rc; }
  *out = p;
  return MPE_OK;
}

int mpe_gg20_pipeline_destroy(mpe_gg20_pipeline* p) {
  if (!p) return MPE_E_ARG;
  for (int li = 0; li < p->lanes; ++li) {
    mpe_gg20_pipeline::Lane& L = p->lane[li];
    if (L.st) (void)hipStreamSynchronize(L.st);
    if (L.stage) (void)mpe_gg20_nonces_free(L.stage);
    if (L.blob) { (void)hipMemset(L.blob, 0, L.blob_bytes); (void)hipFree(L.blob); }
    if (L.ctx) (void)mpe_ctx_destroy(L.ctx);
    if (L.st) (void)hipStreamDestroy(L.st);
    mpe::pipe::wipe_bytes(L.seeds, sizeof L.seeds);
    mpe::pipe::wipe_bytes(L.counters, sizeof L.counters);
  }
  for (auto& t : p->ring) { if (t.submitted) (void)hipEventDestroy(t.submitted); if (t.begin) (void)hipEventDestroy(t.begin); if (t.done) (void)hipEventDestroy(t.done); }
  delete p;
  return MPE_OK;
}

int mpe_gg20_pipeline_submit(mpe_gg20_pipeline* p, const int32_t* d_keyset, const mpe_gg20_nonces* nonces, uint32_t* d_r, uint32_t* d_s, int32_t* d_recid,
                             uint32_t* d_R, int32_t* d_status, void* stream, uint64_t* ticket) {
  if (!p || !nonces) return MPE_E_ARG;
  if (p->K->K > 1 && !d_keyset) return MPE_E_ARG;
  // every refusal comes BEFORE a slot and a ticket are taken: a refused call leaves the pipeline as it was
  mpe_gg20_nonces src = *nonces;
  for (int f = 0; f < mpe::smp::NF; ++f)
    if (!*mpe::smp::nonce_field_ptr(&src, f)) { mpe_set_error_msg("gg20 pipeline: a nonce array is NULL"); return MPE_E_ARG; }
  (void)mpe::pipe::poll_open_group(p, nullptr);
  if (p->lane[p->cur].seeded) { mpe_set_error_msg("gg20 pipeline: a group holds seeded or caller-sampled batches, not both (flush between the two forms)"); return MPE_E_ARG; }
  int li = 0, g = 0;
  MPE_TRY(mpe::pipe::open_slot(p, d_r, d_s, d_recid, d_R, d_status, (hipStream_t)stream, &li, &g, ticket));
  mpe_gg20_pipeline::Lane& L = p->lane[li];
  for (int f = 0; f < mpe::smp::NF; ++f) {
    mpe_gg20_nonce_buf* sb = L.stage;
    const size_t w = sb->per_session[f] * (size_t)p->batch;
    const uint32_t* from = *mpe::smp::nonce_field_ptr(&src, f);
    uint32_t* to = const_cast<uint32_t*>(*mpe::smp::nonce_field_ptr(&sb->view, f)) + (size_t)g * w;
    (void)hipMemcpyAsync(to, from, w * 4, hipMemcpyDeviceToDevice, L.st);
  }
  return mpe::pipe::close_slot(p, li, d_keyset);
}

int mpe_gg20_pipeline_submit_seeded(mpe_gg20_pipeline* p, const int32_t* d_keyset, const uint8_t* h_seed32, uint64_t batch_counter, const uint32_t* d_msg,
                                    uint32_t* d_r, uint32_t* d_s, int32_t* d_recid, uint32_t* d_R, int32_t* d_status, void* stream, uint64_t* ticket) {
  if (!p || !h_seed32 || !d_msg || (batch_counter >> 56) != 0) return MPE_E_ARG;
  if (p->group > mpe::smp::MAX_SLOTS) { mpe_set_error_msg("gg20 pipeline: seeded submission needs group <= 16 (one sampler launch per group)"); return MPE_E_ARG; }
  if (p->K->K > 1 && !d_keyset) return MPE_E_ARG;
  (void)mpe::pipe::poll_open_group(p, nullptr);
  if (p->lane[p->cur].filled != p->lane[p->cur].seeded) { mpe_set_error_msg("gg20 pipeline: a group holds seeded or caller-sampled batches, not both (flush between the two forms)"); return MPE_E_ARG; }
  int li = 0, g = 0;
  MPE_TRY(mpe::pipe::open_slot(p, d_r, d_s, d_recid, d_R, d_status, (hipStream_t)stream, &li, &g, ticket));
  mpe_gg20_pipeline::Lane& L = p->lane[li];
  (void)hipMemcpyAsync(const_cast<uint32_t*>(L.stage->view.msg) + (size_t)g * p->batch * 8, d_msg, (size_t)p->batch * 32, hipMemcpyDeviceToDevice, L.st);
  memcpy(L.seeds + 32 * g, h_seed32, 32);
  L.counters[g] = batch_counter;
  L.seeded++;
  return mpe::pipe::close_slot(p, li, d_keyset);
}

int mpe_gg20_pipeline_flush(mpe_gg20_pipeline* p) {
  if (!p) return MPE_E_ARG;
  int rc = MPE_OK;
  for (int k = 0; k < p->lanes; ++k) {
    const int li = (p->cur + k) % p->lanes;
    if (p->lane[li].filled) { const int r = mpe::pipe::launch_group(p, li); if (rc == MPE_OK) rc = r; }
  }
  return rc;
}

int mpe_gg20_pipeline_query(mpe_gg20_pipeline* p, uint64_t ticket, int* done) {
  if (!p || !done) return MPE_E_ARG;
  mpe_gg20_pipeline::Ticket* t = mpe::pipe::ticket_of(p, ticket);
  if (!t) return MPE_E_ARG;
  if (!t->launched) (void)mpe::pipe::poll_open_group(p, nullptr);
  *done = (t->launched && hipEventQuery(t->done) == hipSuccess) ? 1 : 0;
  return (*done && t->rc != MPE_OK) ? t->rc : MPE_OK;           // a completed batch whose pass failed says so
}

int mpe_gg20_pipeline_wait(mpe_gg20_pipeline* p, uint64_t ticket) {
  if (!p) return MPE_E_ARG;
  mpe_gg20_pipeline::Ticket* t = mpe::pipe::ticket_of(p, ticket);
  if (!t) return MPE_E_ARG;
  if (!t->launched) (void)mpe::pipe::launch_group(p, t->lane);          // its group is still open: it goes now, partly filled
  const hipError_t e = hipEventSynchronize(t->done);
  if (e != hipSuccess) { mpe_set_error("gg20 pipeline wait", e); return MPE_E_HIP; }
  if (t->rc != MPE_OK) mpe_set_error_msg("gg20 pipeline: the pass that carried this batch failed (status arrays hold MPE_GG20_STATUS_PASS_FAILED)");
  return t->rc;
}

int mpe_gg20_pipeline_stream_wait(mpe_gg20_pipeline* p, uint64_t ticket, void* stream) {
  if (!p) return MPE_E_ARG;
  mpe_gg20_pipeline::Ticket* t = mpe::pipe::ticket_of(p, ticket);
  if (!t) return MPE_E_ARG;
  if (!t->launched) (void)mpe::pipe::launch_group(p, t->lane);
  const hipError_t e = hipStreamWaitEvent((hipStream_t)stream, t->done, 0);
  if (e != hipSuccess) { mpe_set_error("gg20 pipeline stream wait", e); return MPE_E_HIP; }
  return t->rc;                                                          // known at launch time: the failure is not asynchronous
}

int mpe_gg20_pipeline_latency_ms(mpe_gg20_pipeline* p, uint64_t ticket, float* ms) {
  if (!p || !ms) return MPE_E_ARG;
  mpe_gg20_pipeline::Ticket* t = mpe::pipe::ticket_of(p, ticket);
  if (!t || !t->launched) return MPE_E_ARG;
  hipError_t e = hipEventSynchronize(t->done);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, t->submitted, t->done);
  if (e != hipSuccess) { mpe_set_error("gg20 pipeline latency", e); return MPE_E_HIP; }
  return t->rc;
}

int mpe_gg20_pipeline_pass_ms(mpe_gg20_pipeline* p, uint64_t ticket, float* ms) {
  if (!p || !ms) return MPE_E_ARG;
  mpe_gg20_pipeline::Ticket* t = mpe::pipe::ticket_of(p, ticket);
  if (!t || !t->launched) return MPE_E_ARG;
  mpe_gg20_pipeline::Ticket* t0 = mpe::pipe::ticket_of(p, t->begin_of);
  if (!t0) return MPE_E_ARG;
  hipError_t e = hipEventSynchronize(t->done);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, t0->begin, t->done);
  if (e != hipSuccess) { mpe_set_error("gg20 pipeline pass time", e); return MPE_E_HIP; }
  return t->rc;
}

int mpe_gg20_pipeline_ticket_rc(mpe_gg20_pipeline* p, uint64_t ticket, int* launched, int* rc) {
  if (!p || !rc) return MPE_E_ARG;
  mpe_gg20_pipeline::Ticket* t = mpe::pipe::ticket_of(p, ticket);
  if (!t) return MPE_E_ARG;
  if (launched) *launched = t->launched ? 1 : 0;
  *rc = t->launched ? t->rc : MPE_OK;
  return MPE_OK;
}

int mpe_gg20_pipeline_set_deadline_us(mpe_gg20_pipeline* p, int64_t us) {
  if (!p) return MPE_E_ARG;
  p->deadline_us = us < 0 ? -1 : us;
  return MPE_OK;
}
int mpe_gg20_pipeline_set_eager(mpe_gg20_pipeline* p, int on) {
  if (!p) return MPE_E_ARG;
  p->eager = on ? 1 : 0;
  return MPE_OK;
}
int mpe_gg20_pipeline_poll(mpe_gg20_pipeline* p, int* launched) {
  if (!p) return MPE_E_ARG;
  if (launched) *launched = 0;
  (void)mpe::pipe::poll_open_group(p, launched);                // a failed pass is its tickets' news
  return MPE_OK;
}
int mpe_gg20_pipeline_inject_fault(mpe_gg20_pipeline* p, int passes, int rc) {
  if (!p || passes < 0 || (rc != MPE_E_NOMEM && rc != MPE_E_HIP)) return MPE_E_ARG;
  p->fail_next = passes;
  p->fail_rc = rc;
  return MPE_OK;
}
int mpe_gg20_pipeline_counters(const mpe_gg20_pipeline* p, uint64_t* groups, uint64_t* by_deadline, uint64_t* by_idle, uint64_t* failed) {
  if (!p) return MPE_E_ARG;
  if (groups) *groups = p->launched_groups;
  if (by_deadline) *by_deadline = p->launched_by_deadline;
  if (by_idle) *by_idle = p->launched_by_idle;
  if (failed) *failed = p->failed_groups;
  return MPE_OK;
}

int mpe_gg20_pipeline_sampler_failures(mpe_gg20_pipeline* p, int32_t* h_out) {
  if (!p || !h_out) return MPE_E_ARG;
  int32_t tot = 0;
  for (int li = 0; li < p->lanes; ++li) {
    int32_t v = 0;
    (void)hipStreamSynchronize(p->lane[li].st);
    if (hipMemcpy(&v, p->lane[li].fail, 4, hipMemcpyDeviceToHost) != hipSuccess) return MPE_E_HIP;
    tot += v;
  }
  *h_out = tot;
  return MPE_OK;
}

}  // extern "C"
