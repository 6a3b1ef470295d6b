"""Multi-GPU layouts of the GG20 signing path.  One process per GPU, `torch.distributed` — backend "nccl"
IS RCCL on ROCm (xGMI inside a node); the same code runs under "gloo" on CPU (tests/test_dist_cpu.py).

Mode A, session-sharded (`shard_range`): sessions are independent units; every GPU runs whole sessions
(`mpe_gg20_sign`), the small key tables are replicated, NO data-path collective.

Mode B, party-sharded (`PartySharded`): the parties of a session live on DIFFERENT GPUs and every round's messages travel
through ONE all-gather, after which each party filters what is addressed to it — exactly what the reference's relay does
(every message, P2P ones included, is broadcast to the room and filtered by the client: examples/gg20_sm_client.rs:35-40;
the state machine it feeds: src/protocols/multi_party_ecdsa/gg_2020/state_machine/sign.rs:252-438).  Per round and
rank the payload is the fixed-size record slab of the round engine (include/mpecdsa_hip.h "GG20 round messages"),
so the collective is a single `all_gather_into_tensor` of equal-size blocks: bandwidth-optimal on the point-to-point xGMI
mesh, no ring all-reduce anywhere.  Placements:
  * "party":   world == a divisor pattern of S (rank r hosts the signer ordinals p with p % world == r) — one party per
               rank when world == S: a rank only ever holds its own party's secrets;
  * "rotated": the sessions are cut into `world` blocks and party p of block s lives on rank (s + p) % world — every
               rank hosts S (block, party) pairs, so any world size is perfectly balanced (config 5: t=2, n=5, S=3 on 8
               GPUs); with world >= S no two parties of a session share a GPU.  world < S would put two parties of a
               session on one rank: refused unless the caller says `colocate=True` (a throughput run on fewer GPUs than
               signers; each party's secrets still live in their own key object).

The exchange is stream-ordered: a round's records are written by `mpe_gg20_roundN` straight into this rank's slot of the
gather buffer, the all-gather is queued behind them on the same stream (RCCL) and the next round reads the gathered buffer
in place through `h_in_off` — no host synchronisation, no staging copy, two alternating buffers.  Under gloo (CPU tensors,
or GPU engines that share one device in the tests) the slab is staged through host memory.

The fan-out itself lives BEHIND THE C-ABI (round 5): `mpe_comm_*` / `mpe_gg20_round_exchange` / `mpe_gg20_shard_*` of
include/mpecdsa_hip.h (csrc/mpe_comm.h: ncclAllGather on the round's stream, the placement arithmetic, the layout self-test) — what a
compiled host binds (include/mpecdsa.hpp: PartySharded).  This module is the Python harness around the same entry points: pass
`comm=engine.Comm(...)` and every collective of `run()` is `mpe_comm_all_gather` ("native" gather mode); the placement always comes
from `mpe_gg20_shard_where`.  The torch.distributed modes below remain for the CPU tests (gloo, oracle engines) and for ranks that
share one device (RCCL refuses two ranks on one GPU).

Gather modes (`PartySharded.gather_mode`; MPE_DIST_GATHER in the environment forces one): "native" — the C-ABI communicator;
"inplace" — torch.distributed on RCCL:
the collective's input is this rank's slice of its output; "outofplace" — the input is a separate copy of that slice (what
`layout_self_test` falls back to if the in-place form ever misplaces a row on some backend / version); "staged" — synchronise,
copy through host memory (gloo with GPU engines; also the conservative fallback).  `layout_self_test()` runs one collective of
known row patterns on the REAL backend before any timing and picks the first mode under which every row of every rank lands
where `in_off` says it does — all ranks agree on the mode through an all-reduce."""
import os
import time

import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous, balanced shard [lo, hi) of `total` independent units for `rank` of `world`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(value, device="cpu"):
    """Largest `value` (e.g. elapsed seconds) over all ranks: the job finishes when its slowest rank does."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_tables(tables, device, src=0):
    """Key tables built on rank `src` (dict name -> int32/uint32 torch tensor, identical shapes known to every rank
    through `tables` templates) broadcast once; returns the dict on `device`."""
    out = {}
    for name in sorted(tables):
        t = tables[name].to(device).contiguous()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(t, src=src)
        out[name] = t
    return out


ROUNDS_OUT = (0, 1, 2, 3, 4, 5, 7)        # rounds that emit a message


class PartySharded:
    """Party-sharded GG20 signing over `world` ranks.

    make_engine(block, local_parties) -> an object with
        round(rnd, d_in, in_off, msg[, out]) -> tensor [len(local_parties), Bblk, W] (or None for rounds 6 and 8)
        result() -> dict of tensors (status, bad_actors, r, s, recid [L, Bblk, ..])
    d_in is the gathered slab of the previous round (a flat tensor of records) and in_off[j] the record offset of sender
    ordinal j's [Bblk][W] block in it — the calling convention of mpe_gg20_roundN.  An engine that sets
    `writes_in_place = True` accepts `out=` (a contiguous [L, Bblk, W] view of the gather buffer) and writes its records
    there; the others return a tensor that is copied into the slot."""

    def __init__(self, S, Bblk, msg_words, make_engine, device, placement="rotated", rank=None, world=None, colocate=False,
                 timing=False, comm=None):
        import multi_party_ecdsa_amd.engine as _E        # the placement arithmetic is the library's (host code: no GPU needed)
        self.comm = comm                                 # engine.Comm: the collectives go through mpe_comm_all_gather
        self.S, self.Bblk, self.msg_words, self.device = S, Bblk, msg_words, torch.device(device)
        self.dist = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank() if self.dist else 0) if rank is None else rank
        self.world = (dist.get_world_size() if self.dist else 1) if world is None else world
        G = self.world
        if placement == "party":
            if S % G:
                raise ValueError("placement 'party' needs the world size to divide the number of signers")
            self.blocks = 1
            self._where = lambda s, p: _E.shard_where(_E.PLACE_PARTY, S, G, s, p)
        elif placement == "rotated":
            if G < S and not colocate:
                raise ValueError(f"placement 'rotated' with world {G} < {S} signers puts two parties of a session on one rank "
                                 "(pass colocate=True to accept that)")
            self.blocks = G
            self._where = lambda s, p: _E.shard_where(_E.PLACE_ROTATED, S, G, s, p)
        else:
            raise ValueError(placement)
        self.placement = placement
        hosted = [(s, p) for s in range(self.blocks) for p in range(S) if self._where(s, p)[0] == self.rank]
        self.per_rank = len(hosted)                                   # the same on every rank by construction
        # one engine per block; its local parties in ascending order; slot of each (block, party) in this rank's slab
        self.engines = {}
        for s in sorted({s for s, _ in hosted}):
            parties = sorted(p for ss, p in hosted if ss == s)
            self.engines[s] = (parties, make_engine(s, parties))
        self.comm_s = 0.0
        self.bytes_per_round = {}
        self._bufs = [None, None]
        self._events = []
        self.timing = timing                   # record a HIP event pair around every RCCL all-gather (bench.py); off for a service loop
        self.backend = dist.get_backend() if self.dist else None
        self.maxw = max(self.msg_words(r) for r in ROUNDS_OUT)
        cuda = self.device.type == "cuda"
        self.gather_mode = "inplace" if (cuda and self.backend == "nccl") else ("staged" if cuda else "outofplace")
        forced = os.environ.get("MPE_DIST_GATHER")
        if forced in ("inplace", "outofplace", "staged") and cuda:
            self.gather_mode = forced
        if comm is not None:
            self.gather_mode = "native"
        self.self_test = None

    def in_off(self, s):
        """record offset, in the gathered slab, of every sender ordinal's block for session block s"""
        off = []
        for j in range(self.S):
            r, slot = self._where(s, j)
            off.append((r * self.per_rank + slot) * self.Bblk)
        return off

    def _buffer(self, q, W):
        """gather buffer q & 1 viewed as [world * per_rank, Bblk, W] (rank r's slab = rows [r * per_rank, (r+1) * per_rank))"""
        rows = self.world * self.per_rank
        if self._bufs[q & 1] is None:
            self._bufs[q & 1] = torch.empty(rows * self.Bblk * self.maxw, dtype=torch.int32, device=self.device)
        return self._bufs[q & 1][: rows * self.Bblk * W].view(rows, self.Bblk, W)

    def _collective(self, buf, mine, mode):
        if mode == "native":                         # ncclAllGather behind the C-ABI, queued on the current stream
            self.comm.all_gather(buf, mine.numel() * 4)
        elif mode == "inplace":                      # input = this rank's slice of the output
            dist.all_gather_into_tensor(buf.view(-1), mine.reshape(-1))
        elif mode == "outofplace":
            dist.all_gather_into_tensor(buf.view(-1), mine.clone().reshape(-1))
        else:                                        # staged through host memory
            h_out = torch.empty(buf.shape, dtype=buf.dtype)
            dist.all_gather_into_tensor(h_out.view(-1), mine.cpu().reshape(-1))
            buf.copy_(h_out)

    def _gather(self, buf, mine):
        """all ranks' slabs into `buf` (`mine` = this rank's rows of it, already written)"""
        cuda = buf.is_cuda
        if self.comm is None and (not self.dist or (self.world == 1 and not (cuda and self.backend == "nccl"))):
            return                                   # (one RCCL rank still issues the collective: the same call path as N ranks)
        mode = self.gather_mode if cuda else "outofplace"
        if cuda and mode != "staged":
            # RCCL all-gather queued behind the round's kernels on the current stream; timed with events when asked
            if self.timing:
                if len(self._events) >= 4096:        # a caller that never drains: keep the newest, bounded
                    self._events = self._events[-1024:]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._collective(buf, mine, mode)
                e1.record()
                self._events.append((e0, e1))
            else:
                self._collective(buf, mine, mode)
            return
        if cuda:
            torch.cuda.synchronize(buf.device)       # the wait for the round's kernels is not communication time
        t0 = time.perf_counter()
        self._collective(buf, mine, mode)
        self.comm_s += time.perf_counter() - t0

    def layout_self_test(self):
        """One all-gather of known patterns on the real backend, BEFORE any timed or signing work: row (rank r, slot k) carries
        the value (r * per_rank + k) * 65536 + column; after the collective every row of every rank must sit where `in_off`
        will look for it.  Tries the current mode first, then the remaining ones; all ranks settle on the first mode that is
        right everywhere (all-reduce MIN of the verdicts).  Returns and stores {"mode", "ok", "tried": {mode: ok}}."""
        if getattr(self, "comm", None) is not None:  # mpe_comm_layout_self_test: the same pattern test on the C-ABI communicator
            st = self.comm.layout_self_test(self.per_rank)
            self.self_test = dict(mode="native:" + st["mode"], ok=st["ok"], tried={"native:" + st["mode"]: st["ok"]})
            return self.self_test
        if not self.dist:
            self.self_test = dict(mode=self.gather_mode, ok=True, tried={}, skipped="no process group")
            return self.self_test
        cuda = self.device.type == "cuda"
        rows, W = self.world * self.per_rank, 16
        cols = torch.arange(self.Bblk * W, dtype=torch.int32, device=self.device).view(1, self.Bblk, W) % 65536
        ids = torch.arange(rows, dtype=torch.int32, device=self.device).view(rows, 1, 1)
        want = ids * 65536 + cols
        order = [self.gather_mode] + [m for m in (("inplace", "outofplace", "staged") if cuda else ("outofplace",)) if m != self.gather_mode]
        if cuda and self.backend != "nccl":
            order = ["staged"]
        tried = {}
        for mode in order:
            buf = torch.full((rows, self.Bblk, W), -1, dtype=torch.int32, device=self.device)
            mine = buf[self.rank * self.per_rank:(self.rank + 1) * self.per_rank]
            mine.copy_(want[self.rank * self.per_rank:(self.rank + 1) * self.per_rank])
            try:
                self._collective(buf, mine, mode)
                if cuda:
                    torch.cuda.synchronize(self.device)
                ok = bool(torch.equal(buf, want))
            except RuntimeError:
                ok = False
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if (cuda and self.backend == "nccl") else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            tried[mode] = bool(flag.item())
            if tried[mode]:
                if cuda:
                    self.gather_mode = mode
                self.self_test = dict(mode=mode, ok=True, tried=tried)
                return self.self_test
        self.self_test = dict(mode=None, ok=False, tried=tried)
        raise RuntimeError(f"all-gather layout self-test failed under every mode: {tried}")

    def comm_seconds(self):
        """time spent in the all-gathers since the last call (drains the event pairs: call it after a synchronize)"""
        for e0, e1 in self._events:
            self.comm_s += e0.elapsed_time(e1) * 1e-3
        self._events = []
        t, self.comm_s = self.comm_s, 0.0
        return t

    def run(self, msgs):
        """msgs: {block: tensor [Bblk, 8]} for the blocks this rank hosts.  Returns {block: result dict}."""
        if not self.timing:
            self._events = []
        gathered, q = None, 0
        for rnd in range(9):
            W = self.msg_words(rnd) if rnd in ROUNDS_OUT else 0
            buf = self._buffer(q, W) if W else None
            mine = buf[self.rank * self.per_rank:(self.rank + 1) * self.per_rank] if W else None
            for s, (parties, eng) in self.engines.items():
                slots = [self._where(s, p)[1] for p in parties]
                msg = msgs[s] if rnd == 7 else None
                if W and getattr(eng, "writes_in_place", False) and slots == list(range(slots[0], slots[0] + len(slots))):
                    eng.round(rnd, gathered, self.in_off(s), msg, out=mine[slots[0]:slots[0] + len(slots)])
                else:
                    out = eng.round(rnd, gathered, self.in_off(s), msg)
                    if out is not None:
                        for li, slot in enumerate(slots):
                            mine[slot] = out[li]
            if W:
                self.bytes_per_round[rnd] = buf.numel() * 4
                self._gather(buf, mine)
                gathered = buf.view(-1)
                q += 1
        return {s: eng.result() for s, (parties, eng) in self.engines.items()}
