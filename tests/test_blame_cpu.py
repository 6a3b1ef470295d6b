"""Identifiable abort on the oracle (oracle/gg20_oracle.c: orc_gg20_blame5/6/7 = gg_2020/blame.rs) with the reference's own
fault-injection matrix (gg_2020/test.rs:69-148): the chosen parties double their delta_i (step 5), sigma_i (step 6) or s_i
(step 7); the failing check is the one the reference reaches (R_dash sum -> 502, S_i sum -> 602, signature -> 701) and the
blame function names EXACTLY the corrupted set.  The C blame functions are pinned by the Python restatement
(tests/pyref_gg20.py: blame5 / blame6 / blame7) on the same openings."""
import numpy as np
import pytest

import fixtures as F
import gg20_fixture as G
import orc
import pyref
import pyref_gg20 as PG

# the reference's cases: (t, n, signers, corrupt_step, corrupted signer ordinals)
MATRIX = [(1, 2, [0, 1], 5, [0]), (1, 2, [0, 1], 5, [1]), (1, 2, [0, 1], 5, [0, 1]), (1, 2, [0, 1], 6, [0]), (1, 2, [0, 1], 6, [0, 1]),
          (1, 2, [0, 1], 7, [1]), (2, 5, [0, 2, 3, 4], 5, [0, 3]), (2, 5, [0, 2, 3, 4], 6, [0]), (2, 5, [0, 2, 3, 4], 7, [1, 3])]
EXPECT_STATUS = {5: 502, 6: 602, 7: 701}


def run_oracle_with_faults(lk, nonces, B, step, corrupted):
    S = lk["S"]
    parties = [G.OracleParty(lk, i, B, G.party_nonces(nonces, lk, i)) for i in range(S)]
    for i in corrupted:
        parties[i].fault(step)
    slabs = G.run_rounds(parties, nonces["msg"])
    return parties, slabs


def openings6(lk, nonces, slabs, parties, B, ecddh_nonce):
    """LocalStatePhase6 of every signer computed on the oracle: miu / miu_randomness through Paillier::open of the incoming
    w_i ciphertexts, the ECDDH proof from the party's sigma_i"""
    S, keys = lk["S"], lk["keys"]
    sg = [int(x) for x in lk["arrays"]["signers"]]
    P1 = S - 1
    cb = G.blame6_cb(lk, slabs, B)
    pw, qw = F.words([k.p for k in keys], 32), F.words([k.q for k in keys], 32)
    kidx = np.array([sg[(r // P1) % S] for r in range(B * S * P1)], dtype=np.int32)
    miu, mr = orc.u32((B * S * P1, 64)), orc.u32((B * S * P1, 64))
    orc.lib.orc_paillier_open(B * S * P1, len(keys), orc._p(pw), orc._p(qw), orc._p(kidx), orc._p(cb), orc._p(miu), orc._p(mr))
    Rr = parties[0].result()["R"]
    Svec = np.ascontiguousarray(np.transpose(slabs[5][:, :, 0:16], (1, 0, 2)).reshape(B * S, 16))
    a1, a2, z = orc.u32((B * S, 16)), orc.u32((B * S, 16)), orc.u32((B * S, 8))
    Gw = F.point_words([pyref.G])
    for i in range(S):
        sig = orc.u32((B, 8))
        orc.lib.orc_gg20_party_sigma(parties[i].h, orc._p(sig))
        h1 = orc.ec_mul_base(sig)
        g1 = np.repeat(Gw, B, axis=0)
        o1, o2, oz = orc.u32((B, 16)), orc.u32((B, 16)), orc.u32((B, 8))
        h2 = np.ascontiguousarray(Svec.reshape(B, S, 16)[:, i])
        orc.lib.orc_ecddh_prove(B, orc._p(sig), orc._p(np.ascontiguousarray(ecddh_nonce.reshape(B, S, 8)[:, i])), orc._p(g1), orc._p(h1), orc._p(Rr),
                                orc._p(h2), orc._p(o1), orc._p(o2), orc._p(oz))
        a1.reshape(B, S, 16)[:, i], a2.reshape(B, S, 16)[:, i], z.reshape(B, S, 8)[:, i] = o1, o2, oz
    n = lk["n"]
    return dict(k=nonces["k"].copy(), k_rand=nonces["r_a"].copy(), miu=miu, miu_rand=mr, a1=a1, a2=a2, z=z, S=Svec,
                c_a=np.ascontiguousarray(np.transpose(slabs[0][:, :, n * 256:n * 256 + 128], (1, 0, 2)).reshape(B * S, 128)), c_b=cb, R=Rr)


def openings7(lk, nonces, slabs, parties, B):
    S = lk["S"]
    Rr = parties[0].result()["R"]
    r = F.words([x % pyref.Q for x in F.ints(Rr[:, :8])], 8)
    tr = lambda a: np.ascontiguousarray(np.transpose(a, (1, 0, 2)).reshape(B * S, a.shape[2]))
    return dict(s=tr(slabs[7]), r=r, R_dash=tr(slabs[4][:, :, 450 * (S - 1):450 * (S - 1) + 16]), m=nonces["msg"].copy(), R=Rr, S=tr(slabs[5][:, :, 0:16]))


def py_blame(lk, which, o, b):
    """the Python restatement on the openings of session b"""
    S, keys = lk["S"], lk["keys"]
    sg = [int(x) for x in lk["arrays"]["signers"]]
    P1 = S - 1
    N = [keys[a].N for a in sg]
    row = lambda f, w: F.ints(np.ascontiguousarray(o[f].reshape(-1, w)))
    per = lambda vals, per_item=1: [vals[(b * S + i) * per_item:(b * S + i + 1) * per_item] for i in range(S)]
    one = lambda vals: [v[0] for v in per(vals)]
    pts = lambda f: [v[0] for v in per(F.points(o[f]))]
    if which == "b5":
        return PG.blame5(N, one(row("k", 8)), one(row("k_rand", 64)), one(row("gamma", 8)), per(row("beta_tag", 64), P1), per(row("beta_rand", 64), P1),
                         one(row("delta", 8)), pts("g_gamma"), one(row("c_a", 128)), per(row("c_b", 128), P1))
    if which == "b6":
        X = F.points(lk["arrays"]["X"])
        g_w = [pyref.ec_mul(PG.lagrange(sg, i), X[sg[i]]) for i in range(S)]
        proofs = list(zip(pts("a1"), pts("a2"), one(row("z", 8))))
        return PG.blame6(N, g_w, one(row("k", 8)), one(row("k_rand", 64)), per(row("miu", 64), P1), per(row("miu_rand", 64), P1), proofs, pts("S"),
                         one(row("c_a", 128)), per(row("c_b", 128), P1), F.points(o["R"])[b])
    return PG.blame7(one(row("s", 8)), F.ints(o["r"])[b], pts("R_dash"), F.ints(o["m"])[b], F.points(o["R"])[b], pts("S"))


@pytest.mark.parametrize("t,n,signers,step,corrupted", MATRIX)
def test_fault_injection_names_exactly_the_corrupted_set(keys, t, n, signers, step, corrupted):
    B = 1
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed=f"blame-{t}-{n}-{step}-{corrupted}")
    parties, slabs = run_oracle_with_faults(lk, nonces, B, step, corrupted)
    for p in parties:
        assert list(p.result()["status"]) == [EXPECT_STATUS[step]] * B
    if step == 5:
        which, o = "b5", G.blame5_opened(lk, nonces, slabs, B)
    elif step == 6:
        en = F.words([F.Rng("ecddh").below(pyref.Q - 1) + 1 for _ in range(B * lk["S"])], 8)
        which, o = "b6", openings6(lk, nonces, slabs, parties, B, en)
    else:
        which, o = "b7", openings7(lk, nonces, slabs, parties, B)
    want = sum(1 << i for i in corrupted)
    assert list(G.oracle_blame(lk, which, o, B)) == [want] * B
    if len(signers) == 2:                                   # the Python restatement (small shape: it is slow)
        assert py_blame(lk, which, o, 0) == sorted(corrupted)


def test_blame5_on_a_wrong_opening_blames_the_liar(keys):
    """A signer that opens a k_i different from the one it encrypted in MessageA is named (blame.rs:128-138), and then nobody
    else is examined (the reference's `if bad_signers_vec.is_empty()`)"""
    B = 1
    lk = G.make_local_keys(keys, 1, 3, [0, 2])
    nonces = G.make_nonces(lk, B, seed="blame-liar")
    parties, slabs = run_oracle_with_faults(lk, nonces, B, 5, [0])
    o = G.blame5_opened(lk, nonces, slabs, B)
    o["k"][1, 0] ^= 2
    assert list(G.oracle_blame(lk, "b5", o, B)) == [0b10]
    assert py_blame(lk, "b5", o, 0) == [1]


def test_paillier_open_recovers_the_randomness(keys):
    r = F.Rng("open")
    k = keys[3]
    m, rr = r.below(k.N), r.below(k.N)
    c = pyref.paillier_encrypt(k.N, m, rr)
    assert PG.paillier_open(k.p, k.q, c) == (m, rr)
    mo, ro = orc.u32((1, 64)), orc.u32((1, 64))
    orc.lib.orc_paillier_open(1, 1, orc._p(F.words([k.p], 32)), orc._p(F.words([k.q], 32)), None, orc._p(F.words([c], 128)), orc._p(mo), orc._p(ro))
    assert (F.ints(mo)[0], F.ints(ro)[0]) == (m, rr)
