"""CPU test of the N>1 path with the gloo backend, world_size 2: session sharding covers the job exactly once,
and the whole-job timing/throughput reduction (max over ranks, counts gathered) behaves as bench.py assumes.
The per-rank "work" here is the CPU oracle on each rank's shard: the union of the shards' outputs must equal the
single-process result (no data-path collective is needed or used)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("mpe_dist", os.path.join(ROOT, "multi_party_ecdsa_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)          # loaded by path: the package import needs the HIP library, this helper does not
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
    counts = D.gather_counts(hi - lo)
    q.put((rank, lo, hi, F.ints(out), elapsed, counts))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    world, total, port = 2, 11, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures as F
    r = F.Rng("dist")
    mods = [r.bits(2048) | (1 << 2047) | 1 for _ in range(3)]
    base = [r.bits(2048) for _ in range(total)]
    exp = [r.bits(256) for _ in range(total)]
    want = [pow(b, e, mods[i % 3]) for i, (b, e) in enumerate(zip(base, exp))]
    covered = []
    for rank, lo, hi, out, elapsed, counts in res:
        assert out == want[lo:hi]
        covered += list(range(lo, hi))
        assert elapsed == 1.25                             # max over ranks (0.25, 1.25)
        assert counts == [6, 5] and sum(counts) == total
    assert covered == list(range(total))                   # every unit exactly once, in order


def test_shard_range_properties():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mpe_dist", os.path.join(ROOT, "multi_party_ecdsa_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    for total in (0, 1, 7, 65536):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
