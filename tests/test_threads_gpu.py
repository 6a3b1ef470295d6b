"""-m gpu: the threading contract of SURVEY.md §8(b) — "callable from multiple host threads with distinct mpe_ctx* / HIP
streams; no global state".  The reference runs every round through `spawn_blocking` because `is_expensive()` is true
(state_machine/sign/rounds.rs:106,215,323,408,504,598,642) and its `Simulation` harness runs the parties concurrently
(state_machine/sign.rs:667-691).  Here: host threads, one context and one stream each, running AT THE SAME TIME — two
`mpe_gg20_sign` batches over different key sets and signer sets, the round-by-round API on a third, a Paillier encrypt /
decrypt loop on a fourth — every result byte-identical to the oracle.  ctypes releases the GIL around every C call, so the
host sides really overlap.  No entry point reads the environment after a context is created (mpe_lib.hip)."""
import threading

import numpy as np
import pytest
import torch

import fixtures as F
import gg20_fixture as G
import orc

pytestmark = pytest.mark.gpu


def _dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def test_concurrent_contexts_on_distinct_streams(keys):
    from multi_party_ecdsa_amd import engine as E
    from test_gg20_gpu import GpuParty
    jobs = [dict(kind="sign", t=1, n=3, signers=[0, 1], B=12, seed="thr-a", shift=0),
            dict(kind="sign", t=2, n=5, signers=[0, 2, 4], B=6, seed="thr-b", shift=5, share=4),      # mpe_ctx_set_device_share: other lane layouts, same bytes
            dict(kind="rounds", t=1, n=3, signers=[1, 2], B=5, seed="thr-c", shift=9),
            dict(kind="paillier", B=96, seed="thr-d")]
    want = []
    for j in jobs:                                                   # oracle first, single-threaded
        if j["kind"] == "paillier":
            r = F.Rng(j["seed"])
            ks = keys[:4]
            idx = [i % 4 for i in range(j["B"])]
            m = [r.below(ks[i].N) for i in idx]
            rr = [r.coprime_below(ks[i].N) for i in idx]
            c = orc.paillier_encrypt(F.words([k.N for k in ks], 64), F.words(m, 64), F.words(rr, 64), idx)
            j.update(ks=ks, idx=idx, m=m, rr=rr)
            want.append(c)
        else:
            kk = keys[j["shift"]:] + keys[:j["shift"]]               # a different wallet per job
            lk = G.make_local_keys(kk, j["t"], j["n"], j["signers"], seed=j["seed"])
            nonces = G.make_nonces(lk, j["B"], seed=j["seed"])
            j.update(lk=lk, nonces=nonces)
            want.append(G.oracle_sign_ex(lk, nonces, j["B"]))
    got, errors = [None] * len(jobs), []
    start = threading.Barrier(len(jobs))

    def work(ix):
        j = jobs[ix]
        try:
            ctx = E.Context(0)
            if j.get("share"):
                ctx.set_device_share(j["share"])
            with torch.cuda.stream(torch.cuda.Stream(device=ctx.device)):          # torch's current stream is per thread
                assert ctx.stream().value != 0
                start.wait(timeout=120)
                for _rep in range(2):                                              # keep the contexts busy together
                    if j["kind"] == "paillier":
                        ks = j["ks"]
                        sk = E.PaillierKeys(ctx, p=[k.p for k in ks], q=[k.q for k in ks])
                        di = torch.tensor(j["idx"], dtype=torch.int32, device=ctx.device)
                        c = sk.encrypt_device(_dev(ctx, F.words(j["m"], 64)), _dev(ctx, F.words(j["rr"], 64)), di)
                        m = sk.decrypt_device(c, di)
                        ctx.sync()
                        got[ix] = (_u32(c), F.ints(_u32(m)))
                    elif j["kind"] == "sign":
                        gk = E.Gg20Keys(ctx, j["t"], j["n"], j["signers"], j["lk"]["arrays"])
                        r, s, recid, status, R = E.gg20_sign(ctx, gk, {f: _dev(ctx, v) for f, v in j["nonces"].items()}, j["B"], want_R=True)
                        ctx.sync()
                        got[ix] = (_u32(r), _u32(s), recid.cpu().numpy(), status.cpu().numpy(), _u32(R))
                    else:
                        S = len(j["signers"])
                        gk = E.Gg20Keys(ctx, j["t"], j["n"], j["signers"], j["lk"]["arrays"])
                        p = GpuParty(ctx, gk, j["B"], list(range(S)), j["nonces"])
                        slabs, prev = {}, None
                        for rnd in range(9):
                            out = p.round(rnd, j["nonces"]["msg"] if rnd == 7 else prev)
                            if rnd in G.ROUNDS:
                                slabs[rnd] = prev = out
                        got[ix] = (slabs, p.result())
        except Exception as e:                                                     # noqa: BLE001
            errors.append((ix, repr(e)))
            try:
                start.abort()
            except Exception:                                                      # noqa: BLE001
                pass

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors
    assert all(not th.is_alive() for th in threads)
    for ix, (j, w) in enumerate(zip(jobs, want)):
        g = got[ix]
        if j["kind"] == "paillier":
            assert np.array_equal(g[0], w) and g[1] == j["m"]
        elif j["kind"] == "sign":
            assert not g[3].any() and not w["status"].any()
            assert np.array_equal(g[0], w["r"]) and np.array_equal(g[1], w["s"]) and list(g[2]) == list(w["recid"]) and np.array_equal(g[4], w["R"])
        else:
            for rnd in G.ROUNDS:
                assert np.array_equal(g[0][rnd], w["slabs"][rnd]), f"round {rnd}"
            assert not g[1]["status"].any() and np.array_equal(g[1]["r"][0], w["r"])


def test_last_error_is_per_thread(gpu_ctx):
    """mpe_last_error() is thread-local: an argument error raised on one thread does not show up on another"""
    from multi_party_ecdsa_amd import _native as N
    seen = {}

    def bad():
        N.lib.mpe_ctx_set_encoding(gpu_ctx.h, N.Encoding())                       # all-zero orders: not permutations
        seen["bad"] = N.lib.mpe_last_error().decode()

    def clean():
        seen["clean"] = N.lib.mpe_last_error().decode()
    t1 = threading.Thread(target=bad); t1.start(); t1.join()
    t2 = threading.Thread(target=clean); t2.start(); t2.join()
    assert "permutation" in seen["bad"] and seen["clean"] == ""
