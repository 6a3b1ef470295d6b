"""GPU tests of the N>1 paths on the ONE device a test box has:
 * Mode B with the real round engine: `dist.PartySharded` at world 2 and 3, every rank a separate PROCESS with its own
   `mpe_ctx`, key object (only its parties' secrets) and `mpe_gg20_session`s on cuda:0; the round slabs travel through a
   gloo all-gather (staged through host memory — RCCL refuses two ranks on one device); every rank's `mpe_gg20_roundN`
   reads the gathered buffer in place through `h_in_off` and writes its records into its slot of the next buffer.
   Signatures, R and status of every (block, party) equal the oracle's lock-step run (`orc_gg20_sign_ex`).  This is the
   topology of the reference's deployment: one party per process, messages relayed (examples/gg20_sm_client.rs:35-40);
 * `mpe_gg20_round1..complete` fed a slab whose sender blocks are PERMUTED and separated by garbage padding (a non-default
   `h_in_off`): same messages, same signatures as the oracle;
 * `mpe_gg20_session_rearm`: a second batch on the same session objects equals a fresh session's output;
 * `bench.py --gpus 2 --share-device`: the self-spawning N>1 launch produces one JSON line with n_gpus = 2, both modes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import fixtures as F
import gg20_fixture as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


# ---- Mode B, GPU engines, one process per rank ---------------------------------------------------------------------------
def _worker(rank, world, port, t, n, signers, B, placement, q, backend="gloo", runs=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # this pool's driver only supports dmabuf IPC
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(rank)                                   # RCCL: one device per rank
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from multi_party_ecdsa_amd import dist as D
        from multi_party_ecdsa_amd import engine as E
        ctx = E.Context(rank if backend == "nccl" else 0)
        keys = F.load_keys()
        lk = G.make_local_keys(keys, t, n, signers)
        S = len(signers)
        nonces = G.make_nonces(lk, B, seed=f"modeB-gpu-{t}-{n}-{placement}")
        blocks = world if placement == "rotated" else 1
        Bblk = B // blocks

        def block_nonces(s):
            return {f: np.ascontiguousarray(v[s * Bblk * (v.shape[0] // B):(s + 1) * Bblk * (v.shape[0] // B)]) for f, v in nonces.items()}

        class Eng:
            writes_in_place = True

            def __init__(self, s, parties):
                # a key object with ONLY the hosted parties' secrets (the other rows of x, p, q never reach the device)
                self.gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"], own=[signers[p] for p in parties])
                bn = block_nonces(s)
                mine = {}
                for f, v in bn.items():
                    if f == "msg":
                        mine[f] = v
                    else:
                        per = v.shape[0] // (Bblk * S)
                        mine[f] = np.ascontiguousarray(v.reshape(Bblk, S, per, v.shape[1])[:, parties].reshape(-1, v.shape[1]))
                self.keep = {f: _dev(ctx, v) for f, v in mine.items()}
                self.sess = E.Gg20Session(ctx, self.gk, Bblk, parties, self.keep)

            def round(self, rnd, d_in, in_off, msg, out=None):
                return self.sess.round(rnd, d_in=d_in, in_off=in_off, msg=msg, out=out)

            def result(self):
                return self.sess.result()

            def rearm(self):
                self.sess.rearm(self.keep)      # the same sampled values again: a test of the buffers, never a deployment pattern

        comm = None
        if backend == "nccl":                   # the fan-out behind the C-ABI: rank 0's RCCL id travels through the process group

            def exchange(ident):
                box = [ident]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            comm = E.Comm(ctx, rank, world, exchange_id=exchange)
        ps = D.PartySharded(S, Bblk, lambda rnd: E.gg20_msg_words(S, n, rnd), Eng, ctx.device, placement=placement, comm=comm)
        selftest = ps.layout_self_test() if backend == "nccl" else None
        msgs = {s: _dev(ctx, block_nonces(s)["msg"]) for s in ps.engines}
        res = ps.run(msgs)
        ctx.sync()
        for _ in range(runs - 1):               # consecutive batches on the same objects and the same double buffers
            first = {s: {f: v.clone() for f, v in r.items()} for s, r in res.items()}
            for eng, _parties in [(e, p) for (p, e) in ps.engines.values()]:
                eng.rearm()
            res = ps.run(msgs)
            ctx.sync()
            for s in res:
                for f in res[s]:
                    assert torch.equal(res[s][f], first[s][f]), f"run after re-arm differs in {f} of block {s}"
        out = {s: {f: (_u32(v) if f in ("r", "s", "R", "bad_actors") else v.cpu().numpy()).tolist() for f, v in r.items()} for s, r in res.items()}
        hosted = {s: parties for s, (parties, _) in ps.engines.items()}
        q.put((rank, hosted, out, dict(ps.bytes_per_round)) + ((selftest,) if selftest is not None else ()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _spawn(world, args, **kw):
    import queue
    import time
    import torch.multiprocessing as mp
    port = 23000 + (os.getpid() * 11 + world * 137 + len(str(args)) * 7) % 4000
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port) + args + (q,), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    res, t0 = [], time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 600:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError(f"worker failed: exit codes {dead}")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda x: x[0])


@pytest.mark.parametrize("world,t,n,signers,B,placement", [
    (2, 1, 3, [0, 2], 4, "party"),            # one party per rank: the reference's deployment
    (2, 1, 3, [1, 2], 6, "rotated"),          # party p of block s on rank (s + p) % 2
    (3, 2, 5, [0, 2, 4], 3, "rotated"),       # BASELINE config 5's shape, three ranks
    # ((3, 2, 4, [0, 1, 3], 2, "party") ran until round 5; the CPU suite keeps that placement at world 3 under gloo)
])
def test_party_sharded_gpu_engines_over_gloo(world, t, n, signers, B, placement):
    res = _spawn(world, (t, n, signers, B, placement), runs=2)       # two batches on the same sessions and gather buffers
    lk = G.make_local_keys(F.load_keys(), t, n, signers)
    nonces = G.make_nonces(lk, B, seed=f"modeB-gpu-{t}-{n}-{placement}")
    want = G.oracle_sign_ex(lk, nonces, B)
    assert not want["status"].any()
    blocks = world if placement == "rotated" else 1
    Bblk = B // blocks
    seen = set()
    for rank, hosted, out, nbytes in res:
        for s, parties in hosted.items():
            r = out[s]
            sl = slice(s * Bblk, (s + 1) * Bblk)
            for li, p in enumerate(parties):
                seen.add((s, p))
                assert r["status"][li] == [0] * Bblk and r["bad_actors"][li] == [0] * Bblk
                assert np.array_equal(np.array(r["r"][li], dtype=np.uint32), want["r"][sl])
                assert np.array_equal(np.array(r["s"][li], dtype=np.uint32), want["s"][sl])
                assert np.array_equal(np.array(r["R"][li], dtype=np.uint32), want["R"][sl])
                assert r["recid"][li] == list(want["recid"][sl])
        assert set(nbytes) == {0, 1, 2, 3, 4, 5, 7}
    assert seen == {(s, p) for s in range(blocks) for p in range(len(signers))}


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two devices: RCCL refuses two ranks on one")
@pytest.mark.parametrize("world,t,n,signers,B,placement", [
    (2, 1, 3, [0, 2], 4, "party"),
    (2, 1, 3, [1, 2], 6, "rotated"),
])
def test_party_sharded_over_rccl_on_distinct_devices(world, t, n, signers, B, placement):
    """the stream-ordered all-gather on the REAL backend at world > 1 (a one-GPU test box skips this; the driver's 8-GPU node
    runs it): layout self-test, then two consecutive run() calls on the same sessions and the same double buffers
    (mpe_gg20_session_rearm in between) — both bit-identical to the oracle's lock-step run"""
    res = _spawn(world, (t, n, signers, B, placement), backend="nccl", runs=2)
    lk = G.make_local_keys(F.load_keys(), t, n, signers)
    want = G.oracle_sign_ex(lk, G.make_nonces(lk, B, seed=f"modeB-gpu-{t}-{n}-{placement}"), B)
    blocks = world if placement == "rotated" else 1
    Bblk = B // blocks
    for rank, hosted, out, nbytes, selftest in res:
        assert selftest["ok"] is True and selftest["mode"] in ("native:inplace", "native:copy"), selftest
        for s, parties in hosted.items():
            sl = slice(s * Bblk, (s + 1) * Bblk)
            for li, p in enumerate(parties):
                assert out[s]["status"][li] == [0] * Bblk
                assert np.array_equal(np.array(out[s]["r"][li], dtype=np.uint32), want["r"][sl])
                assert np.array_equal(np.array(out[s]["s"][li], dtype=np.uint32), want["s"][sl])


def test_party_sharded_through_the_c_abi_communicator_world1(gpu_ctx, keys):
    """dist.PartySharded with comm=engine.Comm: every all-gather of run() is mpe_comm_all_gather (ncclAllGather behind the C-ABI) on a
    real RCCL communicator of one rank — both parties of every session on this GPU (colocate), slabs read in place through
    mpe_gg20_shard_in_off's offsets; signatures equal the oracle's.  The same call path the driver's N-GPU run takes."""
    from multi_party_ecdsa_amd import dist as D
    from multi_party_ecdsa_amd import engine as E
    ctx = gpu_ctx
    t, n, signers, B = 1, 3, [0, 2], 5
    S = len(signers)
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed="native-comm-w1")
    comm = E.Comm(ctx, 0, 1)
    assert comm.layout_self_test(S) == dict(mode="inplace", ok=True) or comm.layout_self_test(S)["ok"]
    made = []

    class Eng:
        writes_in_place = True

        def __init__(self, s, parties):
            self.gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"], own=[signers[p] for p in parties])
            mine = {}
            for f, v in nonces.items():
                if f == "msg":
                    mine[f] = v
                else:
                    per = v.shape[0] // (B * S)
                    mine[f] = np.ascontiguousarray(v.reshape(B, S, per, v.shape[1])[:, parties].reshape(-1, v.shape[1]))
            self.keep = {f: _dev(ctx, v) for f, v in mine.items()}
            self.sess = E.Gg20Session(ctx, self.gk, B, parties, self.keep)
            made.append(self)

        def round(self, rnd, d_in, in_off, msg, out=None):
            return self.sess.round(rnd, d_in=d_in, in_off=in_off, msg=msg, out=out)

        def result(self):
            return self.sess.result()
    ps = D.PartySharded(S, B, lambda rnd: E.gg20_msg_words(S, n, rnd), Eng, ctx.device, placement="rotated", rank=0, world=1, colocate=True, comm=comm)
    assert ps.gather_mode == "native" and ps.layout_self_test()["ok"]
    assert ps.in_off(0) == E.shard_in_off(E.PLACE_ROTATED, S, 1, B, 0) == [0, B]
    res = ps.run({0: _dev(ctx, nonces["msg"])})
    ctx.sync()
    want = G.oracle_sign_ex(lk, nonces, B)
    parties, _ = ps.engines[0]
    for li, p in enumerate(parties):
        assert not res[0]["status"][li].cpu().numpy().any()
        assert np.array_equal(_u32(res[0]["r"][li]), want["r"]) and np.array_equal(_u32(res[0]["s"][li]), want["s"])
    assert set(ps.bytes_per_round) == {0, 1, 2, 3, 4, 5, 7}
    for e in made:
        e.sess.close()
        e.gk.close()
    comm.close()


# ---- non-default h_in_off: permuted sender blocks with garbage between them --------------------------------------------
@pytest.mark.parametrize("t,n,signers,B", [(1, 3, [0, 1], 3), (2, 5, [0, 2, 4], 2)])
def test_rounds_read_a_permuted_padded_slab_through_in_off(gpu_ctx, keys, t, n, signers, B):
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed=f"inoff-{t}-{n}")
    want = G.oracle_sign_ex(lk, nonces, B)
    S = len(signers)
    # one object per party (each only its own secrets); the "relay" lays the senders' blocks out in REVERSE order,
    # each preceded by a different amount of garbage records
    parties = []
    for i in range(S):
        gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"], own=[signers[i]])
        keep = {f: _dev(gpu_ctx, v) for f, v in G.party_nonces(nonces, lk, i).items()}
        parties.append((gk, keep, E.Gg20Session(gpu_ctx, gk, B, [i], keep)))
    gen = torch.Generator(device=gpu_ctx.device)
    gen.manual_seed(5)
    slab, off = None, None
    for rnd in range(9):
        W = E.gg20_msg_words(S, n, rnd) if rnd in G.ROUNDS else 0
        outs = []
        for i, (_, _, sess) in enumerate(parties):
            o = sess.round(rnd, d_in=slab, in_off=off, msg=_dev(gpu_ctx, nonces["msg"]) if rnd == 7 else None)
            outs.append(o)
        if W:
            for i in range(S):
                assert np.array_equal(_u32(outs[i])[0], want["slabs"][rnd][i]), f"round {rnd}, party {i}"
            pads = [3 + 2 * j for j in range(S)]                       # records of garbage in front of each block
            total = sum(pads) + S * B + 5
            slab = torch.randint(-2**31, 2**31 - 1, (total, W), dtype=torch.int32, device=gpu_ctx.device, generator=gen)
            off, pos = [0] * S, 0
            for k, j in enumerate(reversed(range(S))):                 # sender S-1 first
                pos += pads[k]
                off[j] = pos
                slab[pos:pos + B] = outs[j][0]
                pos += B
            slab = slab.reshape(-1)
    for i, (_, _, sess) in enumerate(parties):
        res = sess.result()
        gpu_ctx.sync()
        assert not res["status"].cpu().numpy().any()
        assert np.array_equal(_u32(res["r"])[0], want["r"]) and np.array_equal(_u32(res["s"])[0], want["s"])
        assert list(res["recid"].cpu().numpy()[0]) == list(want["recid"])


# ---- session re-arm -----------------------------------------------------------------------------------------------------
def test_rearmed_session_equals_a_fresh_one(gpu_ctx, keys):
    from multi_party_ecdsa_amd import engine as E
    t, n, signers, B = 1, 3, [0, 2], 3
    lk = G.make_local_keys(keys, t, n, signers)
    S = len(signers)
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])
    n1, n2 = G.make_nonces(lk, B, seed="rearm-1"), G.make_nonces(lk, B, seed="rearm-2")
    w2 = G.oracle_sign_ex(lk, n2, B)

    def run(sess, nn):
        prev = None
        slabs = {}
        for rnd in range(9):
            out = sess.round(rnd, d_in=prev, msg=_dev(gpu_ctx, nn["msg"]) if rnd == 7 else None)
            if out is not None:
                prev = out.reshape(-1)
                slabs[rnd] = _u32(out)
        return slabs, sess.result()
    keep1 = {f: _dev(gpu_ctx, v) for f, v in n1.items()}
    sess = E.Gg20Session(gpu_ctx, gk, B, list(range(S)), keep1)
    run(sess, n1)
    keep2 = {f: _dev(gpu_ctx, v) for f, v in n2.items()}
    sess.rearm(keep2)
    slabs, res = run(sess, n2)
    gpu_ctx.sync()
    for rnd in G.ROUNDS:
        assert np.array_equal(slabs[rnd], w2["slabs"][rnd]), f"round {rnd} after re-arm"
    assert not res["status"].cpu().numpy().any()
    for i in range(S):
        assert np.array_equal(_u32(res["r"])[i], w2["r"]) and np.array_equal(_u32(res["s"])[i], w2["s"])


def test_an_abandoned_batch_can_be_given_up_and_the_object_reused(gpu_ctx, keys):
    """a party that stops after the offline stage (or whose peer vanished) is not left with an object it can only destroy:
    a half-run batch refuses mpe_gg20_session_rearm, mpe_gg20_session_abort wipes it, the next batch equals a fresh session's"""
    from multi_party_ecdsa_amd import engine as E
    t, n, signers, B = 1, 3, [0, 1], 2
    lk = G.make_local_keys(keys, t, n, signers)
    S = len(signers)
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])
    n1, n2 = G.make_nonces(lk, B, seed="abort-1"), G.make_nonces(lk, B, seed="abort-2")
    w2 = G.oracle_sign_ex(lk, n2, B)
    sess = E.Gg20Session(gpu_ctx, gk, B, list(range(S)), {f: _dev(gpu_ctx, v) for f, v in n1.items()})
    prev = None
    for rnd in range(3):                                              # rounds 0..2 only
        prev = sess.round(rnd, d_in=prev).reshape(-1)
    keep2 = {f: _dev(gpu_ctx, v) for f, v in n2.items()}
    with pytest.raises(E.N_.MpeError):
        sess.rearm(keep2)                                             # half-way through and healthy: refused
    sess.abort()
    with pytest.raises(E.N_.MpeError):
        sess.round(3, d_in=prev)                                      # the abandoned batch cannot go on
    sess.rearm(keep2)
    prev = None
    for rnd in range(9):
        out = sess.round(rnd, d_in=prev, msg=_dev(gpu_ctx, n2["msg"]) if rnd == 7 else None)
        if out is not None:
            prev = out.reshape(-1)
    res = sess.result()
    gpu_ctx.sync()
    assert not res["status"].cpu().numpy().any()
    for i in range(S):
        assert np.array_equal(_u32(res["r"])[i], w2["r"]) and np.array_equal(_u32(res["s"])[i], w2["s"])
    sess.close()
    gk.close()


# ---- bench.py --gpus N spawns its own ranks -----------------------------------------------------------------------------
@pytest.mark.parametrize("mode,shape", [("session", []), ("party", ["--t", "2", "--n", "5"])])
def test_bench_self_spawns_two_ranks(mode, shape):
    """(the second case: three signers on two ranks — two parties of a session share a rank, `colocate` — and bench flags that
    are prefixes of the launcher's own options, which must reach bench.py untouched)"""
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--steps", "1", "--warmup", "1",
           "--sessions", "256", "--mode", mode, "--no-configs", "--no-cpu-baseline"] + shape
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["all_sessions_signed"] is True
    assert len(rec["per_rank"]["signatures_per_s"]) == 2 and rec["per_rank"]["all_ranks_signed"]
    assert abs(rec["value"] - 2 * 256 * 1 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]
    if mode == "party":
        assert set(rec["config"]["bytes_all_gathered_per_round"]) == {"0", "1", "2", "3", "4", "5", "7"}


def test_the_drivers_one_command_tells_both_modes_and_checks_itself():
    """`bench.py --gpus 2` exactly as the driver starts it (plus --share-device: one GPU here) prints ONE line that carries the
    session-sharded headline, a per-rank parity sample against the oracle, the all-gather LAYOUT self-test on the real backend,
    and a party-sharded pass at BASELINE config 5's shape (t=2, n=5) with its own rate, all-gather share, bytes per round and
    parity — nothing else is needed from the driver to learn about Mode B."""
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--steps", "1", "--warmup", "1",
           "--sessions", "128", "--no-configs", "--mode-b-sessions", "32", "--mode-b-steps", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["all_sessions_signed"] is True
    par = rec["per_rank"]["parity_vs_oracle"]
    assert par["all_ok"] is True and par["ok_per_rank"] == [True, True] and par["sessions_per_rank"] == 16
    st = rec["rccl"]["all_gather_layout_self_test"]
    assert st["ok"] is True and st["mode"] in ("inplace", "outofplace", "staged") and st["tried"][st["mode"]] is True
    mb = rec["mode_b"]
    assert "error" not in mb, mb
    assert mb["all_sessions_signed"] is True and mb["parity_sample_vs_oracle"] is True and mb["parity_sessions_per_rank"] == 4
    assert mb["blocks"] == 2 and mb["signers"] == 3 and mb["sessions_per_block"] == 32
    assert set(mb["bytes_all_gathered_per_round"]) == {"0", "1", "2", "3", "4", "5", "7"}
    assert 0 <= mb["rccl_time_share"] < 1 and mb["signatures_per_s"] > 0 and len(mb["per_rank_signatures_per_s"]) == 2
    assert abs(mb["signatures_per_s"] - 2 * 32 * 1 / (mb["ms_per_step"] * 1e-3)) < 1e-6 * mb["signatures_per_s"]


_COEXIST = r"""
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT={port!r}, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
order = {order!r}
from multi_party_ecdsa_amd import engine as E
ctx = E.Context(0)
def torch_side():
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.ones(4, device="cuda:0"); dist.all_reduce(x); torch.cuda.synchronize()
    assert x.tolist() == [1.0] * 4
def mpe_side():
    comm = E.Comm(ctx, 0, 1)
    st = comm.layout_self_test(2)
    buf = torch.arange(1024, dtype=torch.int32, device="cuda:0")
    comm.all_gather(buf, 4096); ctx.sync()
    assert st["ok"] and buf.cpu().tolist() == list(range(1024))
    return comm
if order == "torch_first":
    torch_side(); comm = mpe_side()
else:
    comm = mpe_side(); torch_side()
# both stay usable side by side
y = torch.full((8,), 2.0, device="cuda:0"); dist.all_reduce(y); torch.cuda.synchronize()
comm.all_gather(torch.zeros(256, dtype=torch.int32, device="cuda:0"), 1024); ctx.sync()
mapped = sorted({{ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln}})
print("RESULT " + json.dumps(dict(lib=E.comm_library(), mapped=mapped, torch_dir=os.path.dirname(torch.__file__))))
comm.close(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("order", ["torch_first", "mpe_first"])
def test_the_c_abi_communicator_shares_rccl_with_torch_distributed(order):
    """First contact with a node where torch.distributed's nccl group and mpe_comm_* live in ONE process (bench.py --mode party under
    torchrun): the library binds RCCL at run time and must ADOPT the copy the process already holds — PyTorch loads its own
    librccl.so with `import torch` — so that the process maps exactly one librccl and both communicators work side by side, in either
    order of creation.  Two copies would mean two sets of proxy threads and IPC handles for the same device; fail loudly."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _COEXIST.format(root=root, port=str(29650 + (order == "mpe_first")), order=order)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert len(res["mapped"]) == 1, f"two RCCL copies in one process: {res['mapped']}"
    assert res["lib"]["adopted"] is True and os.path.realpath(res["lib"]["path"]) == os.path.realpath(res["mapped"][0])
    assert res["lib"]["version"] > 0
