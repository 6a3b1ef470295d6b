"""CPU tests of the N>1 paths with the gloo backend (multi_party_ecdsa_amd/dist.py is engine-agnostic; here the per-party
round engine is the CPU oracle, on the GPU it is mpe_gg20_roundN — same calling convention):
 * Mode A (session sharding, world 2): the shards cover the job exactly once, the job time is the max over ranks;
 * Mode B (party sharding, worlds 2 and 3): every rank holds ONLY its party's secrets, each round's messages travel through
   one all-gather and are filtered by the receiver like the reference's relay (examples/gg20_sm_client.rs:35-40); every
   rank's signatures equal the single-process oracle's; the balanced "rotated" placement (party p of block s on rank
   (s + p) % world) is checked too."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_dist():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mpe_dist", os.path.join(ROOT, "multi_party_ecdsa_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)          # loaded by path: the package import needs the HIP library, this helper does not
    return D


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_a(rank, world, port, total, q):
    _init(rank, world, port)
    D = _load_dist()
    import fixtures as F
    import orc
    lo, hi = D.shard_range(total, rank, world)
    r = F.Rng("dist")                    # every rank derives the same global inputs, then works on its shard only
    mods = [r.bits(2048) | (1 << 2047) | 1 for _ in range(3)]
    base = [r.bits(2048) for _ in range(total)]
    exp = [r.bits(256) for _ in range(total)]
    idx = [i % 3 for i in range(total)]
    out = orc.modexp(F.words(mods, 64), F.words(base[lo:hi], 64), F.words(exp[lo:hi], 8), idx[lo:hi]) if hi > lo else \
        np.zeros((0, 64), dtype=np.uint32)
    dist.barrier()
    elapsed = D.max_over_ranks(0.25 + rank)                # the slowest rank defines the job time
    q.put((rank, lo, hi, F.ints(out), elapsed))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(worker, world, args):
    port = 29500 + (os.getpid() * 7 + world * 131 + hash(worker.__name__) % 97) % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 900:                  # a worker died (its peers would wait for it forever)
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError(f"worker failed: exit codes {dead}")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda x: x[0])


def test_two_rank_sharding_gloo():
    world, total = 2, 11
    res = _spawn(_worker_a, world, (total,))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures as F
    r = F.Rng("dist")
    mods = [r.bits(2048) | (1 << 2047) | 1 for _ in range(3)]
    base = [r.bits(2048) for _ in range(total)]
    exp = [r.bits(256) for _ in range(total)]
    want = [pow(b, e, mods[i % 3]) for i, (b, e) in enumerate(zip(base, exp))]
    covered = []
    for rank, lo, hi, out, elapsed in res:
        assert out == want[lo:hi]
        covered += list(range(lo, hi))
        assert elapsed == 1.25                             # max over ranks (0.25, 1.25)
    assert covered == list(range(total))                   # every unit exactly once, in order


def test_shard_range_properties():
    D = _load_dist()
    for total in (0, 1, 7, 65536):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- Mode B ---------------------------------------------------------------------------------------------------------
class _OracleEngine:
    """the round-engine interface of dist.PartySharded over the CPU oracle: one OracleParty per local party, each created
    from a key struct that holds only that party's secrets"""

    def __init__(self, lk, nonces_block, parties, Bblk):
        import gg20_fixture as G
        self.G, self.S, self.Bblk = G, lk["S"], Bblk
        self.parties = [G.OracleParty(lk, p, Bblk, G.party_nonces(nonces_block, lk, p)) for p in parties]

    def round(self, rnd, d_in, in_off, msg):
        if rnd == 7:
            outs = [p.round(7, np.ascontiguousarray(msg.numpy().view(np.uint32))) for p in self.parties]
        else:
            slab = None if d_in is None else np.ascontiguousarray(d_in.numpy().view(np.uint32))
            outs = [p.round(rnd, slab, in_off) for p in self.parties]
        if outs[0] is None:
            return None
        return torch.from_numpy(np.stack(outs).view(np.int32))

    def result(self):
        res = [p.result() for p in self.parties]
        return {f: np.stack([r[f] for r in res]) for f in res[0]}


class _OracleEngineInPlace(_OracleEngine):
    """the same engine with the calling convention of the GPU one: it is handed its slot of the gather buffer and writes there"""
    writes_in_place = True

    def round(self, rnd, d_in, in_off, msg, out=None):
        res = super().round(rnd, d_in, in_off, msg)
        if res is None or out is None:
            return res
        assert out.is_contiguous() and out.shape == res.shape
        out.copy_(res)
        return out


def _worker_b(rank, world, port, t, n, signers, B, placement, in_place, q):
    _init(rank, world, port)
    D = _load_dist()
    import fixtures as F
    import gg20_fixture as G
    keys = F.load_keys()
    lk = G.make_local_keys(keys, t, n, signers)
    S = len(signers)
    nonces = G.make_nonces(lk, B, seed=f"modeB-{t}-{n}")          # the global job; every rank slices what its parties own
    blocks = world if placement == "rotated" else 1
    Bblk = B // blocks

    def block_nonces(s):
        out = {}
        for f, v in nonces.items():
            per = v.shape[0] // B
            out[f] = np.ascontiguousarray(v[s * Bblk * per:(s + 1) * Bblk * per])
        return out

    def make_engine(s, parties):
        cls = _OracleEngineInPlace if in_place else _OracleEngine
        return cls(lk, block_nonces(s), parties, Bblk)
    ps = D.PartySharded(S, Bblk, lambda rnd: G.msg_words(S, n, rnd), make_engine, "cpu", placement=placement)
    st = ps.layout_self_test()                     # the gather layout on the real backend (gloo here), before any signing work
    assert st["ok"] and st["mode"] == "outofplace" and st["tried"] == {"outofplace": True}
    msgs = {s: torch.from_numpy(block_nonces(s)["msg"].view(np.int32)) for s in ps.engines}
    res = ps.run(msgs)
    out = {s: {f: v.tolist() for f, v in r.items()} for s, r in res.items()}
    hosted = {s: parties for s, (parties, _) in ps.engines.items()}
    q.put((rank, hosted, out, ps.bytes_per_round))
    dist.barrier()
    dist.destroy_process_group()


def _check_mode_b(world, t, n, signers, B, placement, in_place=False):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures as F
    import gg20_fixture as G
    res = _spawn(_worker_b, world, (t, n, signers, B, placement, in_place))
    lk = G.make_local_keys(F.load_keys(), t, n, signers)
    nonces = G.make_nonces(lk, B, seed=f"modeB-{t}-{n}")
    want = G.oracle_sign_ex(lk, nonces, B)
    assert not want["status"].any()
    blocks = world if placement == "rotated" else 1
    Bblk = B // blocks
    seen = set()
    for rank, hosted, out, nbytes in res:
        for s, parties in hosted.items():
            r = out[s]
            for li, p in enumerate(parties):
                seen.add((s, p))
                sl = slice(s * Bblk, (s + 1) * Bblk)
                assert r["status"][li] == [0] * Bblk
                assert np.array_equal(np.array(r["r"][li], dtype=np.uint32), want["r"][sl])
                assert np.array_equal(np.array(r["s"][li], dtype=np.uint32), want["s"][sl])
                assert r["recid"][li] == list(want["recid"][sl])
        assert set(nbytes) == {0, 1, 2, 3, 4, 5, 7}
    assert seen == {(s, p) for s in range(blocks) for p in range(len(signers))}     # every party of every block exactly once


def test_party_sharded_world2_one_party_per_rank():
    _check_mode_b(2, 1, 3, [0, 2], 2, "party")


def test_party_sharded_world3_one_party_per_rank():
    _check_mode_b(3, 2, 4, [0, 1, 3], 1, "party")


def test_party_sharded_rotated_blocks_world3_two_signers():
    _check_mode_b(3, 1, 3, [1, 2], 3, "rotated")


def test_party_sharded_engines_that_write_into_the_gather_buffer():
    """the GPU engine's calling convention (round(out=slot of the gather buffer), two alternating buffers, in-place all-gather)
    with the oracle as the engine: world 2, rotated blocks"""
    _check_mode_b(2, 1, 3, [0, 1], 4, "rotated", in_place=True)


def test_gather_layout_self_test_catches_a_misplaced_row():
    """the self-test really checks placement: a collective that delivers the ranks' slabs in the wrong order is refused under
    that mode and the next mode is tried (single process, world 2 simulated by patching the collective)"""
    import pytest
    D = _load_dist()

    class Fake(D.PartySharded):
        def __init__(self):
            pass
    ps = Fake()
    ps.dist, ps.world, ps.rank, ps.per_rank, ps.Bblk = True, 2, 0, 3, 5
    ps.device, ps.backend, ps.gather_mode, ps.self_test = torch.device("cpu"), "gloo", "outofplace", None
    calls = []

    def bad_collective(buf, mine, mode):
        calls.append(mode)
        rows = ps.world * ps.per_rank
        cols = torch.arange(ps.Bblk * 16, dtype=torch.int32).view(1, ps.Bblk, 16) % 65536
        ids = torch.arange(rows, dtype=torch.int32).view(rows, 1, 1)
        good = ids * 65536 + cols
        buf.copy_(good.flip(0) if len(calls) == 1 else good)          # first attempt: rows reversed
    ps._collective = bad_collective
    import torch.distributed as dist
    real = dist.all_reduce
    dist.all_reduce = lambda t, op=None: None                          # one process: the verdict is this rank's own
    try:
        with pytest.raises(RuntimeError):
            ps.layout_self_test()                                      # CPU tensors only know one mode: it failed, nothing left
    finally:
        dist.all_reduce = real
    assert calls == ["outofplace"] and ps.self_test["ok"] is False
