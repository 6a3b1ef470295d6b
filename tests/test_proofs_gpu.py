"""GPU parity tests: secp256k1 ops, modinv, DLogProof, AliceProof and PDLwSlackProof through the
C-ABI vs the GMP oracle (byte-identical proofs for identical nonces; cross-verification both ways;
the reference's negative tests)."""
import json
import os

import numpy as np
import pytest
import torch

import fixtures as F
import orc
import pyref

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
H = lambda s: int(s, 16)


def E():
    from multi_party_ecdsa_amd import engine
    return engine


def npw(t):
    return np.ascontiguousarray(t.cpu().numpy().view(np.uint32))


def to_dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


@pytest.fixture(scope="module")
def env(gpu_ctx, keys):
    e = E()
    pk = e.PaillierKeys(gpu_ctx, N=[k.N for k in keys[:4]])
    stm = e.Statements(gpu_ctx, [k.Nt for k in keys[4:7]], [k.h1 for k in keys[4:7]], [k.h2 for k in keys[4:7]])
    tabs = dict(N=F.words([k.N for k in keys[:4]], 64), Nt=F.words([k.Nt for k in keys[4:7]], 64),
                h1=F.words([k.h1 for k in keys[4:7]], 64), h2=F.words([k.h2 for k in keys[4:7]], 64))
    return pk, stm, tabs


def test_ec_ops_vs_oracle(gpu_ctx):
    e = E()
    r = F.Rng("gpu-ec")
    B = 70
    ks = [r.below(pyref.Q) for _ in range(B)]
    ks[0], ks[1], ks[2], ks[3] = 0, 1, pyref.Q - 1, pyref.Q + 7          # edges: infinity, G, -G, reduction mod q
    kw = F.words(ks, 9)                                                     # 9 words: exercises the mod-q reduction
    got = npw(e.ec_mul_base(gpu_ctx, to_dev(gpu_ctx, kw)))
    kred = F.words([k % pyref.Q for k in ks], 8)
    want = orc.ec_mul_base(kred)
    assert np.array_equal(got, want)
    P = want.copy()
    P[0] = want[5]                                                          # avoid infinity as a base for item 0
    k2 = F.words([r.below(pyref.Q) for _ in range(B)], 8)
    got = npw(e.ec_mul(gpu_ctx, to_dev(gpu_ctx, k2), to_dev(gpu_ctx, P)))
    assert np.array_equal(got, orc.ec_mul(k2, P))
    Q = np.roll(P, 1, axis=0)
    Q[4] = P[4]                                                             # doubling through the add path
    Q[6] = F.point_words([pyref.ec_neg(F.points(P[6:7])[0])])[0]            # P + (-P) = infinity
    Q[7] = 0                                                                # P + infinity
    got = npw(e.ec_add(gpu_ctx, to_dev(gpu_ctx, P), to_dev(gpu_ctx, Q)))
    assert np.array_equal(got, orc.ec_add(P, Q))
    # wide scalars: alpha < q^3 (24 words) reduced like Scalar::from(&BigInt)
    al = [r.below(pyref.Q ** 3) for _ in range(8)]
    got = npw(e.ec_mul_base(gpu_ctx, to_dev(gpu_ctx, F.words(al, 24))))
    assert np.array_equal(got, orc.ec_mul_base(F.words([a % pyref.Q for a in al], 8)))


def test_modinv_vs_oracle(gpu_ctx, keys):
    e = E()
    r = F.Rng("gpu-modinv")
    for bits in (2048, 4096):
        k32 = bits // 32
        mods = [keys[0].N ** (bits // 2048), r.bits(bits) | (1 << (bits - 1)) | 1, keys[1].p * 3 if bits == 2048 else keys[1].NN]
        B = 24
        idx = [i % 3 for i in range(B)]
        a = [r.below(mods[i]) for i in idx]
        a[0], a[1] = 1, mods[idx[1]] - 1
        a[2] = 3 * 5 if bits == 2048 else keys[1].p                          # not invertible mod mods[2]
        a[5] = 0
        ms = e.ModSet(gpu_ctx, bits, mods)
        out, ok = e.modinv_device(gpu_ctx, ms, e.dev(gpu_ctx, a, k32), torch.tensor(idx, dtype=torch.int32, device=gpu_ctx.device))
        gpu_ctx.sync()
        w_out, w_ok = orc.modinv(F.words(mods, k32), F.words(a, k32), idx)
        assert list(ok.cpu().numpy()) == list(w_ok)
        assert np.array_equal(npw(out), w_out)
        assert w_ok[2] == 0 and w_ok[5] == 0 and w_ok[0] == 1


def test_dlog_proof(gpu_ctx):
    e = E()
    r = F.Rng("gpu-dlog")
    B = 33
    sk, nonce = F.words([r.below(pyref.Q) for _ in range(B)], 8), F.words([r.below(pyref.Q) for _ in range(B)], 8)
    pk, R, z = e.dlog_prove(gpu_ctx, to_dev(gpu_ctx, sk), to_dev(gpu_ctx, nonce))
    wpk, wR, wz = orc.dlog_prove(sk, nonce)
    assert np.array_equal(npw(pk), wpk) and np.array_equal(npw(R), wR) and np.array_equal(npw(z), wz)
    zz = wz.copy()
    zz[3, 0] ^= 1
    ok = e.dlog_verify(gpu_ctx, pk, R, to_dev(gpu_ctx, zz))
    want = [1] * B
    want[3] = 0
    assert list(ok.cpu().numpy()) == want == list(orc.dlog_verify(wpk, wR, zz))


def _alice_inputs(keys, B, seed):
    r = F.Rng(seed)
    kidx, sidx = [i % 4 for i in range(B)], [(i // 2) % 3 for i in range(B)]
    a = [r.below(pyref.Q) for _ in range(B)]
    rr = [r.below(keys[k].N) for k in kidx]
    c = [pyref.paillier_encrypt(keys[k].N, x, y) for k, x, y in zip(kidx, a, rr)]
    nn = [F.alice_nonces(r, keys[k], keys[4 + s]) for k, s in zip(kidx, sidx)]
    return kidx, sidx, a, rr, c, nn


def test_alice_proof_golden_and_oracle(gpu_ctx, keys, env):
    e = E()
    pk, stm, tabs = env
    B = 21
    kidx, sidx, a, rr, c, nn = _alice_inputs(keys, B, "gpu-alice")
    nw = {f: F.words([n[f] for n in nn], w) for f, w in e.ALICE_NONCE_WORDS.items()}
    di = lambda v: torch.tensor(v, dtype=torch.int32, device=gpu_ctx.device)
    pr = e.alice_generate(gpu_ctx, pk, stm, e.dev(gpu_ctx, a, 8), e.dev(gpu_ctx, c, 128), e.dev(gpu_ctx, rr, 64),
                          {f: to_dev(gpu_ctx, v) for f, v in nw.items()}, di(kidx), di(sidx))
    want = orc.alice_generate(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx, F.words(a, 8), F.words(c, 128),
                              F.words(rr, 64), nw["alpha"], nw["beta"], nw["gamma"], nw["rho"])
    for f in want:
        assert np.array_equal(npw(pr[f]), want[f]), f
    # the GPU proof verifies under the oracle, and the GPU verifier accepts it
    assert list(orc.alice_verify(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx, F.words(c, 128),
                                 {f: npw(v) for f, v in pr.items()})) == [1] * B
    ok = e.alice_verify(gpu_ctx, pk, stm, e.dev(gpu_ctx, c, 128), pr, di(kidx), di(sidx))
    assert list(ok.cpu().numpy()) == [1] * B
    # negatives (mirrors the oracle's accept/reject pattern): tampered s / s2 / ciphertext, s1 > q^3, wrong statement
    bad = {f: v.clone() for f, v in pr.items()}
    bad["s"][0, 0] ^= 1
    bad["s2"][1, 40] ^= 2
    bad["s1"][2] = to_dev(gpu_ctx, F.words([pyref.Q ** 3 + 1], 25))[0]
    bad["z"][3, 5] ^= 1
    c2 = e.dev(gpu_ctx, c, 128)
    c2[4, 7] ^= 16
    sidx2 = list(sidx)
    sidx2[5] = (sidx2[5] + 1) % 3
    ok = e.alice_verify(gpu_ctx, pk, stm, c2, bad, di(kidx), di(sidx2))
    want_ok = orc.alice_verify(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx2, npw(c2), {f: npw(v) for f, v in bad.items()})
    assert list(ok.cpu().numpy()) == list(want_ok)
    assert list(want_ok[:6]) == [0] * 6 and list(want_ok[6:]) == [1] * (B - 6)
    # committed golden vectors (pure-Python generated)
    with open(os.path.join(HERE, "golden", "golden_small.json")) as f:
        gold = json.load(f)["alice"]
    pk2 = e.PaillierKeys(gpu_ctx, N=[keys[g["ek"]].N for g in gold])
    stm2 = e.Statements(gpu_ctx, *[[getattr(keys[g["st"]], f) for g in gold] for f in ("Nt", "h1", "h2")])
    pr = e.alice_generate(gpu_ctx, pk2, stm2, e.dev(gpu_ctx, [H(g["a"]) for g in gold], 8),
                          e.dev(gpu_ctx, [H(g["c"]) for g in gold], 128), e.dev(gpu_ctx, [H(g["r"]) for g in gold], 64),
                          {f: e.dev(gpu_ctx, [H(g["nonces"][f]) for g in gold], w) for f, w in e.ALICE_NONCE_WORDS.items()})
    for f in e.ALICE_PROOF_WORDS:
        assert e.host(pr[f]) == [H(g["proof"][f]) for g in gold], f


def test_alice_verify_on_hostile_proof_values(gpu_ctx, keys, env):
    """A prover may send anything: s = N, 0, N - 1, 2^2048 - 1 (>= N), z = N~ or 0, a ciphertext that is a multiple of
    N, of p, or not reduced.  The GPU verifier must return exactly the oracle's verdicts (all rejections here)."""
    e = E()
    pk, stm, tabs = env
    B = 12
    kidx, sidx, a, rr, c, nn = _alice_inputs(keys, B, "gpu-alice-hostile")
    nw = {f: F.words([n[f] for n in nn], w) for f, w in e.ALICE_NONCE_WORDS.items()}
    di = lambda v: torch.tensor(v, dtype=torch.int32, device=gpu_ctx.device)
    pr = e.alice_generate(gpu_ctx, pk, stm, e.dev(gpu_ctx, a, 8), e.dev(gpu_ctx, c, 128), e.dev(gpu_ctx, rr, 64),
                          {f: to_dev(gpu_ctx, v) for f, v in nw.items()}, di(kidx), di(sidx))
    bad = {f: v.clone() for f, v in pr.items()}
    N = [keys[k].N for k in kidx]
    Nt = [keys[4 + s_].Nt for s_ in sidx]
    row = lambda v, w: to_dev(gpu_ctx, F.words([v], w))[0]
    bad["s"][0] = row(N[0], 64)
    bad["s"][1] = row(0, 64)
    bad["s"][2] = row(N[2] - 1, 64)
    bad["s"][3] = row((1 << 2048) - 1, 64)
    bad["z"][4] = row(Nt[4], 64)
    bad["z"][5] = row(0, 64)
    c2 = e.dev(gpu_ctx, c, 128)
    c2[6] = row(N[6] * 7, 128)
    c2[7] = row(keys[kidx[7]].p * 11, 128)
    c2[8] = row((c[8] + N[8] * N[8]) if (c[8] + N[8] * N[8]).bit_length() <= 4096 else c[8], 128)
    c2[9] = row(0, 128)
    bad["s2"][10] = row((1 << (89 * 32)) - 1, 89)           # every 13-bit fixed-base window at its maximum, top one partial
    bad["s1"][11] = row(pyref.Q ** 3, 25)                    # the boundary of the range check (s1 > q^3 rejects)
    ok = e.alice_verify(gpu_ctx, pk, stm, c2, bad, di(kidx), di(sidx))
    want_ok = orc.alice_verify(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx, npw(c2), {f: npw(v) for f, v in bad.items()})
    assert list(ok.cpu().numpy()) == list(want_ok)
    assert sum(want_ok[:8]) == 0 and sum(want_ok[10:]) == 0


def test_pdl_proof_oracle_and_soundness(gpu_ctx, keys, env):
    """zk_pdl_with_slack/test.rs:11-68 (prove -> verify) and :70-129 (ciphertext of x+1 -> reject)"""
    e = E()
    pk, stm, tabs = env
    r = F.Rng("gpu-pdl")
    B = 19
    kidx, sidx = [i % 4 for i in range(B)], [(i // 3) % 3 for i in range(B)]
    x = [r.below(pyref.Q) for _ in range(B)]
    rr = [r.below(keys[k].N) for k in kidx]
    G = [pyref.ec_mul(r.below(pyref.Q), pyref.G) for _ in range(B)]
    Qp = [pyref.ec_mul(xx, g) for xx, g in zip(x, G)]
    c = [pyref.paillier_encrypt(keys[k].N, xx + (1 if i == 2 else 0), y) for i, (k, xx, y) in enumerate(zip(kidx, x, rr))]
    nn = [F.pdl_nonces(r, keys[k], keys[4 + s]) for k, s in zip(kidx, sidx)]
    nw = {f: F.words([n[f] for n in nn], w) for f, w in e.PDL_NONCE_WORDS.items()}
    di = lambda v: torch.tensor(v, dtype=torch.int32, device=gpu_ctx.device)
    dC, dQ, dG = e.dev(gpu_ctx, c, 128), to_dev(gpu_ctx, F.point_words(Qp)), to_dev(gpu_ctx, F.point_words(G))
    pr = e.pdl_prove(gpu_ctx, pk, stm, dC, dQ, dG, e.dev(gpu_ctx, x, 8), e.dev(gpu_ctx, rr, 64),
                     {f: to_dev(gpu_ctx, v) for f, v in nw.items()}, di(kidx), di(sidx))
    want = orc.pdl_prove(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx, F.words(c, 128), F.point_words(Qp),
                         F.point_words(G), F.words(x, 8), F.words(rr, 64), nw["alpha"], nw["beta"], nw["rho"], nw["gamma"])
    for f in want:
        assert np.array_equal(npw(pr[f]), want[f]), f
    ok = e.pdl_verify(gpu_ctx, pk, stm, dC, dQ, dG, pr, di(kidx), di(sidx))
    exp = [1] * B
    exp[2] = 0                                                            # soundness: x+1 was encrypted
    assert list(ok.cpu().numpy()) == exp
    assert list(orc.pdl_verify(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx, F.words(c, 128), F.point_words(Qp),
                               F.point_words(G), {f: npw(v) for f, v in pr.items()})) == exp
    # tamper every field in turn
    for j, f in enumerate(["z", "u1", "u2", "u3", "s1", "s2", "s3"]):
        bad = {k: v.clone() for k, v in pr.items()}
        bad[f][4 + j, 1] ^= 1
        ok = e.pdl_verify(gpu_ctx, pk, stm, dC, dQ, dG, bad, di(kidx), di(sidx))
        w = orc.pdl_verify(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx, F.words(c, 128), F.point_words(Qp),
                           F.point_words(G), {k: npw(v) for k, v in bad.items()})
        assert list(ok.cpu().numpy()) == list(w), f
        assert w[4 + j] == 0
    # hostile values: s2 = N, 0, 2^2048 - 1; u2 = 0 / N^2 - 1; z = N~; a ciphertext that is a multiple of N (not a unit)
    bad = {k: v.clone() for k, v in pr.items()}
    row = lambda v, w: to_dev(gpu_ctx, F.words([v], w))[0]
    N = [keys[k].N for k in kidx]
    bad["s2"][0] = row(N[0], 64)
    bad["s2"][1] = row(0, 64)
    bad["s2"][3] = row((1 << 2048) - 1, 64)
    bad["u2"][5] = row(0, 128)
    bad["u2"][6] = row(N[6] * N[6] - 1, 128)
    bad["z"][7] = row(keys[4 + sidx[7]].Nt, 64)
    dC2 = dC.clone()
    dC2[8] = row(N[8] * 3, 128)
    ok = e.pdl_verify(gpu_ctx, pk, stm, dC2, dQ, dG, bad, di(kidx), di(sidx))
    w = orc.pdl_verify(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx, npw(dC2), F.point_words(Qp),
                       F.point_words(G), {k: npw(v) for k, v in bad.items()})
    assert list(ok.cpu().numpy()) == list(w)
    assert sum(w[i] for i in (0, 1, 3, 5, 6, 7, 8)) == 0


def test_config3_full_size(gpu_ctx, keys):
    """BASELINE config 3 at its full size: 262 144 fixed- and variable-base scalar multiplications and
    262 144 PDLwSlackProof::prove + verify over 16 (ek, N~, h1, h2) tuples: every honest proof accepted, the 1 % corrupted
    ones (a different field each) all rejected and only those, a 4 096-instance prefix bit-exact vs the oracle (the same
    function bench.py reports as configs.c3_ec_pdl_262144)."""
    import bench
    res = bench.config3(gpu_ctx, E(), keys, F, B=262144, prefix=4096)
    assert res["accepted"] == 262144 and res["corrupted_1pct_all_rejected"] and 2600 <= res["corrupted"] <= 2640
    assert res["parity_prefix_4096"] and res["ec_parity_prefix"]


@pytest.mark.parametrize("check", [False, True])
def test_bob_proof_vs_oracle(gpu_ctx, keys, env, check):
    """range_proofs.rs:636-709: generate(check=false) -> verify(None); generate(check=true) -> BobProofExt::verify"""
    e = E()
    pk, stm, tabs = env
    r = F.Rng(f"gpu-bob-{check}")
    B = 11
    kidx, sidx = [i % 4 for i in range(B)], [(i // 2) % 3 for i in range(B)]
    a_enc, mta, bs, bps, rs, nns = [], [], [], [], [], []
    for k, s in zip(kidx, sidx):
        ek, st = keys[k], keys[4 + s]
        a, b, bp, rr = r.below(pyref.Q), r.below(pyref.Q), r.below(ek.N), r.below(ek.N)
        ae = pyref.paillier_encrypt(ek.N, a, r.below(ek.N))
        a_enc.append(ae); bs.append(b); bps.append(bp); rs.append(rr)
        mta.append(pow(ae, b, ek.NN) * pyref.paillier_encrypt(ek.N, bp, rr) % ek.NN)
        nns.append(F.bob_nonces(r, ek, st))
    nw = {f: F.words([n[f] for n in nns], w) for f, w in e.BOB_NONCE_WORDS.items()}
    di = lambda v: torch.tensor(v, dtype=torch.int32, device=gpu_ctx.device)
    dA, dM = e.dev(gpu_ctx, a_enc, 128), e.dev(gpu_ctx, mta, 128)
    pr, u = e.bob_generate(gpu_ctx, pk, stm, dA, dM, e.dev(gpu_ctx, bs, 8), e.dev(gpu_ctx, bps, 64), e.dev(gpu_ctx, rs, 64),
                           {f: to_dev(gpu_ctx, v) for f, v in nw.items()}, check, di(kidx), di(sidx))
    want, wu = orc.bob_generate(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx, F.words(a_enc, 128), F.words(mta, 128),
                                F.words(bs, 8), F.words(bps, 64), F.words(rs, 64), nw["alpha"], nw["beta"], nw["gamma"], nw["rho"],
                                nw["rho_prim"], nw["sigma"], nw["tau"], check)
    for f in want:
        assert np.array_equal(npw(pr[f]), want[f]), f
    X = None
    if check:
        assert np.array_equal(npw(u), wu)
        X = to_dev(gpu_ctx, orc.ec_mul_base(F.words(bs, 8)))
    ok = e.bob_verify(gpu_ctx, pk, stm, dA, dM, pr, X, u, di(kidx), di(sidx))
    assert list(ok.cpu().numpy()) == [1] * B
    # tamper one field per item; GPU verdicts must equal the oracle's
    bad = {k: v.clone() for k, v in pr.items()}
    for j, f in enumerate(["t", "z", "e", "s", "s1", "s2", "t1", "t2"]):
        bad[f][j, 0] ^= 2
    ok = e.bob_verify(gpu_ctx, pk, stm, dA, dM, bad, X, u, di(kidx), di(sidx))
    w_ok = orc.bob_verify(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], kidx, sidx, F.words(a_enc, 128), F.words(mta, 128),
                          {k: npw(v) for k, v in bad.items()}, npw(X) if check else None, npw(u) if check else None)
    assert list(ok.cpu().numpy()) == list(w_ok)
    assert list(w_ok[:8]) == [0] * 8 and list(w_ok[8:]) == [1] * (B - 8)
