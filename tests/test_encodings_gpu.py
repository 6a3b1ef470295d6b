"""-m gpu: the HIP engine under every encoding profile (`mpe_ctx_set_encoding`, include/mpecdsa_hip.h).  The byte-level
conventions of curv-kzen 0.9 / zk-paillier 0.4.3 that the reference's source does not fix — DigestExt::chain_point's point
form, BigInt::to_bytes(0), the order of the points in each sigma proof's challenge, zk-paillier's salt / mask blocks /
CompositeDLogProof field order — are run-time properties of a context.  For every profile of enc_profiles.PROFILES the engine
must equal the C oracle AND the independent Python restatement bit for bit (proofs, verdicts, every round message of whole
signing sessions, the blame openings), and a proof made under another profile must be rejected."""
import numpy as np
import pytest
import torch

import enc_cases as EC
import enc_profiles as ENCS
import fixtures as F
import gg20_fixture as G
import orc
import pyref

pytestmark = pytest.mark.gpu


def _dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.fixture(scope="module")
def enc_ctx():
    """one context whose profile the tests switch (the profile is read per call: no object caches encoded state)"""
    from multi_party_ecdsa_amd import engine as E
    return E.Context(0)


def test_default_profile_and_argument_checks(enc_ctx):
    from multi_party_ecdsa_amd import _native as N
    assert enc_ctx.encoding() == ENCS.DEFAULT.as_dict()
    for bad in (dict(chain_point=2), dict(zero_bytes=7), dict(ord_heg=[0, 1, 2, 3, 4, 5, 5]), dict(ord_dlog=[0, 1, 3]), dict(reserved=1)):
        with pytest.raises(N.MpeError):
            enc_ctx.set_encoding(dict(ENCS.DEFAULT.as_dict(), **bad))
    assert enc_ctx.encoding() == ENCS.DEFAULT.as_dict()                    # a refused profile changes nothing


@pytest.mark.parametrize("name", list(ENCS.PROFILES))
def test_engine_equals_oracle_and_python_under_every_profile(enc_ctx, keys, name):
    inp = EC.inputs(keys)
    prof = ENCS.PROFILES[name]
    enc_ctx.set_encoding(prof.as_dict())
    try:
        got = EC.engine_outputs(enc_ctx, inp)
        with ENCS.applied(prof):
            want_py, want_orc = EC.python_outputs(inp), EC.oracle_outputs(inp)
            assert got == want_py == want_orc
            assert all(EC.oracle_verdicts(inp, got).values())
        assert all(EC.engine_verdicts(enc_ctx, inp, want_py).values())
    finally:
        enc_ctx.set_encoding(ENCS.DEFAULT.as_dict())


def test_each_switch_reaches_its_kernels(enc_ctx, keys):
    """proofs made under the defaults, verified by a context on another profile: exactly the transcripts that profile touches
    are rejected — the same sets as the oracle's and the Python restatement's (tests/test_encodings_cpu.py)"""
    inp = EC.inputs(keys)
    dflt = EC.python_outputs(inp)
    expect_rejected = {"compressed": {"dlog", "pedersen", "heg", "ecddh"}, "zero-empty": {"ck"}, "mask-be": {"ck"},
                       "reordered": {"dlog", "pedersen", "heg", "ecddh", "cdlog"}, "all-alt": {"dlog", "pedersen", "heg", "ecddh", "ck", "cdlog"}}
    try:
        for name, rejected in expect_rejected.items():
            enc_ctx.set_encoding(ENCS.PROFILES[name].as_dict())
            v = EC.engine_verdicts(enc_ctx, inp, dflt)
            assert {k for k, ok in v.items() if not ok} == rejected, name
    finally:
        enc_ctx.set_encoding(ENCS.DEFAULT.as_dict())


@pytest.mark.parametrize("name,t,n,signers,B", [("all-alt", 1, 3, [0, 2], 2), ("reordered", 2, 5, [0, 2, 4], 1), ("compressed", 1, 3, [0, 1, 2], 1)])
def test_whole_signing_sessions_under_an_alternative_profile(keys, name, t, n, signers, B):
    """every round message of every party byte-identical to the per-party oracle under the same profile (all parties in one
    object, then one object per party exchanging slabs), the signature equal to the default one"""
    from multi_party_ecdsa_amd import engine as E
    from test_gg20_gpu import GpuParty
    prof = ENCS.PROFILES[name]
    ctx = E.Context(0, encoding=prof.as_dict())
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed=f"enc-gpu-{name}")
    base = G.oracle_sign_ex(lk, nonces, B)
    with ENCS.applied(prof):
        want = G.oracle_sign_ex(lk, nonces, B)
    assert not want["status"].any() and np.array_equal(want["r"], base["r"]) and np.array_equal(want["s"], base["s"])
    assert [rnd for rnd in G.ROUNDS if not np.array_equal(want["slabs"][rnd], base["slabs"][rnd])] == [1, 2, 5]
    S = len(signers)
    gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"])
    allp = GpuParty(ctx, gk, B, list(range(S)), nonces)
    prev = None
    for rnd in range(9):
        out = allp.round(rnd, nonces["msg"] if rnd == 7 else prev)
        if rnd in G.ROUNDS:
            assert np.array_equal(out, want["slabs"][rnd]), f"message slab of round {rnd} under {name}"
            prev = out
    res = allp.result()
    assert not res["status"].any()
    for i in range(S):
        assert np.array_equal(res["r"][i], want["r"]) and np.array_equal(res["s"][i], want["s"])
    # one object per party
    class One:
        def __init__(self, i):
            self.gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"], own=[signers[i]])
            self.p = GpuParty(ctx, self.gk, B, [i], G.party_nonces(nonces, lk, i))
        def round(self, rnd, slab):
            o = self.p.round(rnd, slab)
            return None if o is None else o[0]
    slabs = G.run_rounds([One(i) for i in range(S)], nonces["msg"])
    for rnd in G.ROUNDS:
        assert np.array_equal(slabs[rnd], want["slabs"][rnd]), f"round {rnd}, one object per party, under {name}"
    # the lock-step composition (mpe_gg20_sign) under the profile: same signatures
    r_, s_, recid, status = E.gg20_sign(ctx, gk, {f: _dev(ctx, v) for f, v in nonces.items()}, B)
    ctx.sync()
    assert not status.cpu().numpy().any() and np.array_equal(_u32(r_), want["r"]) and np.array_equal(_u32(s_), want["s"])


def test_parties_on_different_profiles_reject_each_other_with_the_oracles_status(gpu_ctx, keys):
    """party 1's round-1 message is replaced by the one it would have sent under "compressed" (same ciphertexts, other Schnorr
    challenges): the default-profile receiver fails MessageB::verify_proofs_get_alpha with 201 — same status and bad-actor
    arrays from the engine and from the oracle, nobody signs that session"""
    from multi_party_ecdsa_amd import engine as E
    from test_gg20_gpu import GpuParty
    t, n, signers, B = 1, 3, [0, 1], 2
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed="enc-gpu-mixed")
    S = len(signers)
    with ENCS.applied(ENCS.PROFILES["compressed"]):
        alt = G.oracle_sign_ex(lk, nonces, B)

    def tamper(r, slab):
        if r == 1:
            assert not np.array_equal(slab[1, 1], alt["slabs"][1][1, 1])
            slab[1, 1] = alt["slabs"][1][1, 1]                        # session 1 only; session 0 stays clean
    orc_parties = [G.OracleParty(lk, i, B, G.party_nonces(nonces, lk, i)) for i in range(S)]
    G.run_rounds(orc_parties, nonces["msg"], tamper)
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])

    class One:
        def __init__(self, i):
            self.p = GpuParty(gpu_ctx, gk, B, [i], G.party_nonces(nonces, lk, i))
        def round(self, r, slab):
            o = self.p.round(r, slab)
            return None if o is None else o[0]
    gpu_parties = [One(i) for i in range(S)]
    G.run_rounds(gpu_parties, nonces["msg"], tamper)
    for i in range(S):
        w, g = orc_parties[i].result(), gpu_parties[i].p.result()
        assert list(g["status"][0]) == list(w["status"]) and list(g["bad_actors"][0]) == list(w["bad_actors"]), i
        assert w["status"][0] == 0
    assert orc_parties[0].result()["status"][1] == 201


def test_blame6_openings_and_ecddh_under_an_alternative_profile(keys):
    """phase-6 blame (blame.rs:258-421) under "all-alt": the ECDDH proofs a session publishes equal the oracle's under the same
    profile and mpe_gg20_blame6 names the corrupted party"""
    from multi_party_ecdsa_amd import engine as E
    from test_blame_cpu import openings6, run_oracle_with_faults
    prof = ENCS.PROFILES["all-alt"]
    ctx = E.Context(0, encoding=prof.as_dict())
    t, n, signers, B, corrupted = 1, 3, [0, 1], 2, [1]
    S = len(signers)
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed="enc-gpu-blame6")
    mask = sum(1 << i for i in corrupted)
    en = F.words([F.Rng("enc-ecddh").below(pyref.Q - 1) + 1 for _ in range(B * S)], 8)
    with ENCS.applied(prof):
        oparties, oslabs = run_oracle_with_faults(lk, nonces, B, 6, corrupted)
        o = openings6(lk, nonces, oslabs, oparties, B, en)
        want_mask = list(G.oracle_blame(lk, "b6", o, B))
    gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"])
    dn = {f: _dev(ctx, v) for f, v in nonces.items()}
    sess = E.Gg20Session(ctx, gk, B, list(range(S)), dn)
    sess.fault_inject(6, mask)
    prev = None
    for rnd in range(9):
        out = sess.round(rnd, d_in=prev, msg=dn["msg"] if rnd == 7 else None)
        if rnd in G.ROUNDS:
            assert np.array_equal(_u32(out), oslabs[rnd]), f"round {rnd}"
            prev = out
    miu, a1, a2, z = sess.blame6_state(_dev(ctx, en))
    tr = lambda x: np.ascontiguousarray(np.moveaxis(_u32(x), 0, 1)).reshape(-1, x.shape[-1])
    assert np.array_equal(tr(a1), o["a1"]) and np.array_equal(tr(a2), o["a2"]) and np.array_equal(tr(z), o["z"])
    got = E.gg20_blame6(ctx, gk, B, {f: _dev(ctx, v) for f, v in o.items()})
    ctx.sync()
    assert list(_u32(got)) == want_mask == [mask] * B
    # the same openings judged by a context on the DEFAULT profile: every ECDDH proof fails, every signer is blamed
    dctx = E.Context(0)
    gk2 = E.Gg20Keys(dctx, t, n, signers, lk["arrays"])
    got2 = E.gg20_blame6(dctx, gk2, B, {f: _dev(dctx, v) for f, v in o.items()})
    dctx.sync()
    assert list(_u32(got2)) == list(G.oracle_blame(lk, "b6", o, B))          # the oracle is back on its defaults here
