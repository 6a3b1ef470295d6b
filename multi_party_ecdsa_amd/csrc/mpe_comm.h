// The per-round message fan-out of party-sharded signing behind the C-ABI: RCCL over xGMI (BASELINE config 5).
// In the reference every message — P2P ones included — is broadcast to the room and filtered by the receiving client
// (examples/gg20_sm_client.rs:35-40; the state machine that consumes them: state_machine/sign.rs:252-438).  Here the parties of a
// session live on different GPUs, a round's records are written by mpe_gg20_roundN straight into this rank's slot of a gather
// buffer, ONE ncclAllGather per round — queued behind the round's kernels on the same stream, equal-size blocks, in place —
// delivers every rank's slab, and the next round reads the gathered buffer in place through h_in_off.  One large collective per
// round is the right shape for the point-to-point xGMI mesh (the largest round is 4 KB per party and session: 268 MB per rank at
// 65 536 sessions, a few ms per hop against a multi-second step).  No sub-communicators: all ranks take part in every gather.
// Placement (the same two as multi_party_ecdsa_amd/dist.py, which now calls these entry points):
//   MPE_PLACE_PARTY    party p on rank p % world, world divides S, one session block;
//   MPE_PLACE_ROTATED  the sessions are cut into `world` blocks, party p of block s on rank (s + p) % world: every rank hosts S
//                      (block, party) pairs for any world size; with world >= S no two parties of a session share a GPU.
// Rank r's slab = rows [r * per_rank, (r + 1) * per_rank) of the gather buffer, a row = one (block, party) pair's [batch][W] records.
// Included by mpe_lib.hip.
//
// RCCL is bound at RUN time (dlopen), the first time a communicator entry point is called: a single-GPU user of the library — and
// every host-only helper (mpe_gg20_shard_*) — needs no librccl at all, and a process that already holds a copy (PyTorch ships its
// own librccl.so and loads it with torch.distributed) gets THAT copy: two RCCL instances in one process would each own a set of
// proxy threads and IPC handles for the same devices.  <rccl/rccl.h> is included for its types only; nothing links against it.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "mpe_gg20.h"

struct mpe_comm {
  mpe_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int mode = 0;                    // 0: in place (the input is this rank's slice of the output); 1: from a copy of the slice
  void* copy = nullptr;            // mode 1 staging
  size_t copy_bytes = 0;
  bool tested = false;
};

namespace mpe {
namespace cm {

struct Rccl {
  void* handle = nullptr;
  bool adopted = false;            // the copy was already in the process (RTLD_NOLOAD found it)
  std::string path, error;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};
// nullptr + the error message set when no RCCL can be found
static Rccl* rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* nm : names) {                             // a copy the process already holds wins (PyTorch's own)
      R.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
      if (R.handle) { R.adopted = true; break; }
    }
    if (!R.handle) {
      std::string tried;
      const char* rocm = getenv("ROCM_PATH");                  // a search path, not a code-path switch
      const std::string dirs[] = {"", std::string(rocm ? rocm : "/opt/rocm") + "/lib/"};
      for (const std::string& d : dirs) {
        for (const char* nm : names) {
          R.handle = dlopen((d + nm).c_str(), RTLD_NOW | RTLD_LOCAL);
          if (R.handle) break;
          tried += std::string(tried.empty() ? "" : "; ") + dlerror();
        }
        if (R.handle) break;
      }
      if (!R.handle) { R.error = "mpe_comm: no RCCL in this process and none could be loaded (" + tried + ")"; return; }
    }
    auto sym = [&](const char* n) -> void* {
      void* f = dlsym(R.handle, n);
      if (!f && R.error.empty()) R.error = std::string("mpe_comm: librccl lacks ") + n;
      return f;
    };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
    R.AllReduce = (decltype(R.AllReduce))sym("ncclAllReduce");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    R.GetVersion = (decltype(R.GetVersion))sym("ncclGetVersion");
    Dl_info di;
    if (R.AllGather && dladdr((void*)R.AllGather, &di) && di.dli_fname) R.path = di.dli_fname;
  });
  if (!R.error.empty()) { mpe_set_error_msg(R.error.c_str()); return nullptr; }
  return &R;
}

static int nccl_fail(const char* what, ncclResult_t r) {
  Rccl* R = rccl();
  mpe_set_error_msg((std::string(what) + ": " + (R ? R->GetErrorString(r) : "no RCCL")).c_str());
  return MPE_E_HIP;
}
__global__ void pattern_kernel(uint32_t* buf, int rows_per_rank, int rank, int cols) {       // row id * 65536 + column, this rank's rows only
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows_per_rank * cols) return;
  const int row = rank * rows_per_rank + g / cols;
  buf[(size_t)row * cols + g % cols] = (uint32_t)row * 65536u + (uint32_t)(g % cols);
}
__global__ void pattern_check_kernel(const uint32_t* buf, int rows, int cols, int32_t* bad) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * cols) return;
  if (buf[g] != (uint32_t)(g / cols) * 65536u + (uint32_t)(g % cols)) atomicAdd(bad, 1);
}

static int all_gather(mpe_comm* c, void* d_buf, size_t bytes_per_rank, hipStream_t st) {
  const char* mine = (const char*)d_buf + (size_t)c->rank * bytes_per_rank;
  const void* src = mine;
  if (c->mode == 1) {
    if (bytes_per_rank > c->copy_bytes) {
      if (c->copy) { (void)hipStreamSynchronize(st); (void)hipFree(c->copy); c->copy = nullptr; c->copy_bytes = 0; }
      if (hipMalloc(&c->copy, bytes_per_rank) != hipSuccess) { mpe_set_error_msg("mpe_comm: hipMalloc(staging)"); return MPE_E_NOMEM; }
      c->copy_bytes = bytes_per_rank;
    }
    (void)hipMemcpyAsync(c->copy, mine, bytes_per_rank, hipMemcpyDeviceToDevice, st);
    src = c->copy;
  }
  Rccl* R = rccl();
  if (!R) return MPE_E_HIP;
  const ncclResult_t r = R->AllGather(src, d_buf, bytes_per_rank, ncclUint8, c->comm, st);
  if (r != ncclSuccess) return nccl_fail("ncclAllGather", r);
  return MPE_OK;
}

static int where(int placement, int S, int world, int block, int party, int* rank, int* slot) {
  if (S < 2 || S > 8 || world < 1 || party < 0 || party >= S || block < 0) return MPE_E_ARG;
  if (placement == MPE_PLACE_PARTY) {
    if (S % world || block != 0) return MPE_E_ARG;
    *rank = party % world; *slot = party / world;
    return MPE_OK;
  }
  if (placement == MPE_PLACE_ROTATED) {
    if (block >= world) return MPE_E_ARG;
    *rank = (block + party) % world; *slot = party;      // a rank hosts exactly one block per party ordinal: the ordinal is the slot
    return MPE_OK;
  }
  return MPE_E_ARG;
}

}  // namespace cm
}  // namespace mpe

extern "C" {

int mpe_comm_unique_id(uint8_t* h_id) {
  if (!h_id) return MPE_E_ARG;
  static_assert(MPE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as MPE_COMM_ID_BYTES opaque bytes");
  mpe::cm::Rccl* R = mpe::cm::rccl();
  if (!R) return MPE_E_HIP;
  ncclUniqueId id;
  const ncclResult_t r = R->GetUniqueId(&id);
  if (r != ncclSuccess) return mpe::cm::nccl_fail("ncclGetUniqueId", r);
  memcpy(h_id, id.internal, NCCL_UNIQUE_ID_BYTES);
  return MPE_OK;
}

int mpe_comm_create(mpe_ctx* ctx, const uint8_t* h_id, int rank, int world, mpe_comm** out) {
  if (!ctx || !h_id || !out || world < 1 || rank < 0 || rank >= world) return MPE_E_ARG;
  mpe::cm::Rccl* R = mpe::cm::rccl();
  if (!R) return MPE_E_HIP;
  mpe_comm* c = new (std::nothrow) mpe_comm();
  if (!c) return MPE_E_NOMEM;
  c->ctx = ctx; c->rank = rank; c->world = world;
  (void)hipSetDevice(ctx->device);
  ncclUniqueId id;
  memcpy(id.internal, h_id, NCCL_UNIQUE_ID_BYTES);
  const ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; return mpe::cm::nccl_fail("ncclCommInitRank", r); }
  *out = c;
  return MPE_OK;
}

int mpe_comm_destroy(mpe_comm* c) {
  if (!c) return MPE_E_ARG;
  if (c->comm) if (mpe::cm::Rccl* R = mpe::cm::rccl()) (void)R->CommDestroy(c->comm);
  if (c->copy) (void)hipFree(c->copy);
  delete c;
  return MPE_OK;
}
// Which RCCL the communicator entry points are bound to: its file (dladdr of ncclAllGather), whether that copy was already in
// the process when the library first needed it (*adopted = 1: e.g. PyTorch's), and ncclGetVersion.  Binds RCCL if nothing has yet.
int mpe_comm_library(char* path_buf, size_t path_cap, int* adopted, int* version) {
  mpe::cm::Rccl* R = mpe::cm::rccl();
  if (!R) return MPE_E_HIP;
  if (path_buf && path_cap) { strncpy(path_buf, R->path.c_str(), path_cap - 1); path_buf[path_cap - 1] = 0; }
  if (adopted) *adopted = R->adopted ? 1 : 0;
  if (version) { int v = 0; (void)R->GetVersion(&v); *version = v; }
  return MPE_OK;
}
int mpe_comm_rank(const mpe_comm* c) { return c ? c->rank : MPE_E_ARG; }
int mpe_comm_world(const mpe_comm* c) { return c ? c->world : MPE_E_ARG; }
int mpe_comm_gather_mode(const mpe_comm* c) { return c ? c->mode : MPE_E_ARG; }

int mpe_comm_all_gather(mpe_comm* c, void* d_buf, size_t bytes_per_rank, void* stream) {
  if (!c || !d_buf) return MPE_E_ARG;
  if (bytes_per_rank == 0) return MPE_OK;
  return mpe::cm::all_gather(c, d_buf, bytes_per_rank, (hipStream_t)stream);
}

// One all-gather of known row patterns on the real communicator BEFORE any signing work: row (rank r, slot k) carries
// (r * rows_per_rank + k) * 65536 + column; afterwards every row of every rank must sit where h_in_off will look for it.  The in-place
// form is tried first, then the copy form; the ranks settle on the first form that is right EVERYWHERE (ncclAllReduce MIN of the verdicts).
int mpe_comm_layout_self_test(mpe_comm* c, int rows_per_rank, int* h_mode, int* h_ok, void* stream) {
  if (!c || rows_per_rank < 1 || !h_ok) return MPE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int cols = 1024, rows = rows_per_rank * c->world;
  uint32_t* buf = nullptr;
  int32_t* flag = nullptr;
  if (hipMalloc((void**)&buf, (size_t)rows * cols * 4) != hipSuccess || hipMalloc((void**)&flag, 8) != hipSuccess) {
    if (buf) (void)hipFree(buf);
    mpe_set_error_msg("mpe_comm self-test: hipMalloc");
    return MPE_E_NOMEM;
  }
  int rc = MPE_OK, ok = 0;
  for (int mode = 0; mode < 2 && rc == MPE_OK && !ok; ++mode) {
    c->mode = mode;
    (void)hipMemsetAsync(buf, 0xff, (size_t)rows * cols * 4, st);
    (void)hipMemsetAsync(flag, 0, 8, st);
    hipLaunchKernelGGL(mpe::cm::pattern_kernel, dim3(mpe::blocks_for(rows_per_rank * cols, 256)), dim3(256), 0, st, buf, rows_per_rank, c->rank, cols);
    rc = mpe::cm::all_gather(c, buf, (size_t)rows_per_rank * cols * 4, st);
    if (rc != MPE_OK) break;
    hipLaunchKernelGGL(mpe::cm::pattern_check_kernel, dim3(mpe::blocks_for(rows * cols, 256)), dim3(256), 0, st, buf, rows, cols, flag);
    // every rank's count of misplaced words, summed: zero = right everywhere
    const ncclResult_t r = mpe::cm::rccl()->AllReduce(flag, flag + 1, 1, ncclInt32, ncclSum, c->comm, st);      // (all_gather above succeeded: RCCL is bound)
    if (r != ncclSuccess) { rc = mpe::cm::nccl_fail("ncclAllReduce", r); break; }
    int32_t h[2] = {0, 0};
    if (hipMemcpyAsync(h, flag, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { mpe_set_error_msg("mpe_comm self-test: copy"); rc = MPE_E_HIP; break; }
    ok = h[1] == 0;
  }
  (void)hipFree(buf);
  (void)hipFree(flag);
  if (rc != MPE_OK) return rc;
  c->tested = true;
  if (h_mode) *h_mode = c->mode;
  *h_ok = ok;
  if (!ok) { mpe_set_error_msg("mpe_comm: the all-gather misplaces rows under both forms"); return MPE_E_HIP; }
  return MPE_OK;
}

int mpe_gg20_shard_where(int placement, int n_signers, int world, int block, int party, int* rank, int* slot) {
  if (!rank || !slot) return MPE_E_ARG;
  return mpe::cm::where(placement, n_signers, world, block, party, rank, slot);
}
int mpe_gg20_shard_blocks(int placement, int n_signers, int world) {
  if (placement == MPE_PLACE_PARTY) return (world >= 1 && n_signers % world == 0) ? 1 : MPE_E_ARG;
  if (placement == MPE_PLACE_ROTATED) return world >= 1 ? world : MPE_E_ARG;
  return MPE_E_ARG;
}
int mpe_gg20_shard_per_rank(int placement, int n_signers, int world) {
  if (placement == MPE_PLACE_PARTY) return (world >= 1 && n_signers % world == 0) ? n_signers / world : MPE_E_ARG;
  if (placement == MPE_PLACE_ROTATED) return world >= 1 ? n_signers : MPE_E_ARG;         // world blocks x S parties over world ranks
  return MPE_E_ARG;
}
// record offset, in the gathered slab, of every sender ordinal's [batch][W] block for session block `block`: the h_in_off of mpe_gg20_roundN
int mpe_gg20_shard_in_off(int placement, int n_signers, int world, int batch, int block, int64_t* h_in_off) {
  if (!h_in_off || batch < 1) return MPE_E_ARG;
  const int per_rank = mpe_gg20_shard_per_rank(placement, n_signers, world);
  if (per_rank < 0) return MPE_E_ARG;
  for (int j = 0; j < n_signers; ++j) {
    int r = 0, slot = 0;
    MPE_TRY(mpe::cm::where(placement, n_signers, world, block, j, &r, &slot));
    h_in_off[j] = ((int64_t)r * per_rank + slot) * batch;
  }
  return MPE_OK;
}
// the records of round `round` of every rank: this rank's rows [rank * per_rank, (rank + 1) * per_rank) of d_slab ([world * per_rank]
// [batch][W(round)]) are already written (by mpe_gg20_roundN with d_out pointing into them); ONE all-gather fills the rest
int mpe_gg20_round_exchange(mpe_comm* c, int n_signers, int n, int round, int per_rank, int batch, uint32_t* d_slab, void* stream) {
  if (!c || !d_slab || per_rank < 1 || batch < 1) return MPE_E_ARG;
  const int W = mpe::gg::msg_words(n_signers, n, round);
  if (W <= 0) return MPE_E_ARG;
  return mpe::cm::all_gather(c, d_slab, (size_t)per_rank * batch * W * 4, (hipStream_t)stream);
}

}  // extern "C"
