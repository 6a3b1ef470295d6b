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
               rank hosts S (block, party) pairs, so any world size >= S is perfectly balanced (config 5: t=2, n=5,
               S=3 on 8 GPUs) and no two parties of a session share a GPU."""
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
        round(rnd, d_in, in_off, msg) -> tensor [len(local_parties), Bblk, W] (or None for rounds 6 and 8)
        result() -> dict of tensors (status, bad_actors, r, s, recid [L, Bblk, ..])
    d_in is the gathered slab of the previous round (a flat tensor of records) and in_off[j] the record offset of sender
    ordinal j's [Bblk][W] block in it — the calling convention of mpe_gg20_roundN."""

    def __init__(self, S, Bblk, msg_words, make_engine, device, placement="rotated", rank=None, world=None):
        self.S, self.Bblk, self.msg_words, self.device = S, Bblk, msg_words, device
        self.dist = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank() if self.dist else 0) if rank is None else rank
        self.world = (dist.get_world_size() if self.dist else 1) if world is None else world
        G = self.world
        if placement == "party":
            if S % G:
                raise ValueError("placement 'party' needs the world size to divide the number of signers")
            self.blocks = 1
            self._where = lambda s, p: (p % G, p // G)
        elif placement == "rotated":
            self.blocks = G
            self._where = lambda s, p: ((s + p) % G, p)
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

    def in_off(self, s):
        """record offset, in the gathered slab, of every sender ordinal's block for session block s"""
        off = []
        for j in range(self.S):
            r, slot = self._where(s, j)
            off.append((r * self.per_rank + slot) * self.Bblk)
        return off

    def _gather(self, rank_slab):
        t0 = time.perf_counter()
        if self.dist and self.world > 1:
            out = torch.empty((self.world * rank_slab.shape[0],) + tuple(rank_slab.shape[1:]), dtype=rank_slab.dtype,
                              device=rank_slab.device)                  # rank r's slab = rows [r * per_rank, (r+1) * per_rank)
            if rank_slab.is_cuda:
                torch.cuda.synchronize(rank_slab.device)
                t0 = time.perf_counter()
            dist.all_gather_into_tensor(out, rank_slab)
            if rank_slab.is_cuda:
                torch.cuda.synchronize(rank_slab.device)
        else:
            out = rank_slab
        self.comm_s += time.perf_counter() - t0
        return out.reshape(-1)

    def run(self, msgs):
        """msgs: {block: tensor [Bblk, 8]} for the blocks this rank hosts.  Returns {block: result dict}."""
        S, Bblk = self.S, self.Bblk
        gathered = None
        for rnd in range(9):
            W = self.msg_words(rnd) if rnd in ROUNDS_OUT else 0
            slab = torch.zeros((self.per_rank, Bblk, W), dtype=torch.int32, device=self.device) if W else None
            for s, (parties, eng) in self.engines.items():
                out = eng.round(rnd, gathered, self.in_off(s), msgs[s] if rnd == 7 else None)
                if out is not None:
                    for li, p in enumerate(parties):
                        slab[self._where(s, p)[1]] = out[li]
            if slab is not None:
                self.bytes_per_round[rnd] = slab.numel() * 4 * self.world
                gathered = self._gather(slab)
        return {s: eng.result() for s, (parties, eng) in self.engines.items()}
