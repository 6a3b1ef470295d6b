"""GPU parity test of the batched GG20 signing pipeline (mpe_gg20_sign) against the CPU oracle
(oracle/gg20_oracle.c): byte-identical (r, s, recid) and R for identical nonces, for the reference's own
(t, n, signer-set) cases (state_machine/sign.rs:740-763); signatures re-checked by the independent
Python ECDSA verifier (the reference's check_sig, gg_2020/test.rs:711-748)."""
import numpy as np
import pytest
import torch

import fixtures as F
import gg20_fixture as G
import pyref

pytestmark = pytest.mark.gpu


def _dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


def _run(gpu_ctx, keys, t, n, signers, B, seed, **kw):
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed=seed)
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])
    out = E.gg20_sign(gpu_ctx, gk, {f: _dev(gpu_ctx, v) for f, v in nonces.items()}, B, want_R=True, **kw)
    gpu_ctx.sync()
    return lk, nonces, [o.cpu().numpy() for o in out]


@pytest.mark.parametrize("t,n,signers,B,kw", [
    (1, 3, [0, 1], 5, {}),
    (1, 3, [0, 2], 3, {"dedup_verify": True}),
    (1, 3, [1, 2], 7, {"chunk": 3}),                 # ragged chunking: 3 + 3 + 1
    (2, 5, [0, 2, 4], 3, {}),
    (2, 4, [1, 2, 3], 2, {"dedup_verify": True}),
])
def test_sign_matches_oracle(gpu_ctx, keys, t, n, signers, B, kw):
    lk, nonces, (r, s, recid, status, R) = _run(gpu_ctx, keys, t, n, signers, B, f"gpu-{t}-{n}-{signers}", **kw)
    wr, ws, wrecid, wR, wstatus = G.oracle_sign(lk, nonces, B)
    assert list(wstatus) == [0] * B
    assert list(status) == [0] * B
    assert np.array_equal(r.view(np.uint32), wr) and np.array_equal(s.view(np.uint32), ws)
    assert list(recid) == list(wrecid)
    assert np.array_equal(R.view(np.uint32), wR)
    for b in range(B):
        m = F.ints(nonces["msg"][b:b + 1])[0]
        assert pyref.ecdsa_verify(lk["y"], m, F.ints(wr[b:b + 1])[0], F.ints(ws[b:b + 1])[0])


def test_large_batch_kernels_on_a_small_batch(keys):
    """Small batches take the latency-oriented variants (twice the lanes per exponentiation, lane groups in the EC
    round kernels); MPE_NO_ADAPTIVE_LANES forces the throughput variants the big batches use: same signatures."""
    import os
    from multi_party_ecdsa_amd import engine as E
    old = os.environ.get("MPE_NO_ADAPTIVE_LANES")
    os.environ["MPE_NO_ADAPTIVE_LANES"] = "1"
    try:
        ctx2 = E.Context(0)
    finally:
        if old is None:
            os.environ.pop("MPE_NO_ADAPTIVE_LANES", None)
        else:
            os.environ["MPE_NO_ADAPTIVE_LANES"] = old
    B = 4
    lk, nonces, (r, s, recid, status, R) = _run(ctx2, keys, 2, 5, [0, 1, 3], B, "gpu-throughput-variants")
    wr, ws, wrecid, wR, wstatus = G.oracle_sign(lk, nonces, B)
    assert list(status) == [0] * B == list(wstatus)
    assert np.array_equal(r.view(np.uint32), wr) and np.array_equal(s.view(np.uint32), ws) and list(recid) == list(wrecid)


def test_wrong_public_key_fails_only_that_check(gpu_ctx, keys):
    """phase6_check_S_i_sum: with an inconsistent y every session reports 601, like the oracle's 602 family"""
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    lk["arrays"]["y"][:] = F.point_words([pyref.ec_mul(999, pyref.G)])
    nonces = G.make_nonces(lk, 2, seed="gpu-bad-y")
    gk = E.Gg20Keys(gpu_ctx, 1, 3, [0, 1], lk["arrays"])
    r, s, recid, status = E.gg20_sign(gpu_ctx, gk, {f: _dev(gpu_ctx, v) for f, v in nonces.items()}, 2)
    gpu_ctx.sync()
    assert list(status.cpu().numpy()) == [601, 601]
    assert list(G.oracle_sign(lk, nonces, 2)[4]) == [602, 602]


def test_config4_sessions_byte_identical_to_the_threaded_oracle(gpu_ctx, keys):
    """BASELINE config 4's shape (t=1, n=3, signers {1,2}, one LocalKey, distinct nonces and messages per session) at
    a size the oracle finishes in seconds on the host cores: 192 sessions, every (r, s, recid, R) byte-identical."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    B = 192
    lk, nonces, (r, s, recid, status, R) = _run(gpu_ctx, keys, 1, 3, [0, 1], B, "gpu-config4")
    assert list(status) == [0] * B
    threads = max(1, min(os.cpu_count() or 1, 32))
    chunks = [c for c in np.array_split(np.arange(B), threads) if len(c)]
    outs = {}

    def run(ix):
        outs[int(ix[0])] = (len(ix), G.oracle_sign(lk, nonces, B, first=int(ix[0]), count=len(ix)))
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, chunks))
    for first, (cnt, (wr, ws, wrecid, wR, wstatus)) in outs.items():
        sl = slice(first, first + cnt)
        assert list(wstatus[sl]) == [0] * cnt
        assert np.array_equal(r.view(np.uint32)[sl], wr[sl]) and np.array_equal(s.view(np.uint32)[sl], ws[sl])
        assert list(recid[sl]) == list(wrecid[sl]) and np.array_equal(R.view(np.uint32)[sl], wR[sl])


def test_inconsistent_key_share_is_caught_in_the_same_round(gpu_ctx, keys):
    """A signer whose share x_i does not match its public X_i: Bob's g^{w} check in verify_proofs_get_alpha (rounds.rs:281)
    fails for every session.  GPU and oracle must stop in the same round (hundreds digit of the status)."""
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, 1, 3, [0, 2])
    lk["arrays"]["x"][0, 0] ^= 1                                   # party 1's share off by a bit; X, y untouched
    B = 3
    nonces = G.make_nonces(lk, B, seed="gpu-bad-share")
    gk = E.Gg20Keys(gpu_ctx, 1, 3, [0, 2], lk["arrays"])
    r, s, recid, status = E.gg20_sign(gpu_ctx, gk, {f: _dev(gpu_ctx, v) for f, v in nonces.items()}, B)
    gpu_ctx.sync()
    want = G.oracle_sign(lk, nonces, B)[4]
    got = status.cpu().numpy()
    assert all(int(x) != 0 for x in got) and all(int(x) != 0 for x in want)
    assert [int(x) // 100 for x in got] == [int(x) // 100 for x in want]
