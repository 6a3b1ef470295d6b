"""Keygen verification math on the oracle (NiCorrectKeyProof / CompositeDLogProof verify, Feldman validate_share — gg_2020/
party_i.rs:260-438) against the Python restatement, on honest and on broken proofs."""
import numpy as np

import fixtures as F
import keygen_fixture as KF
import orc
import pyref


def test_correct_key_proof(keys):
    ks = keys[:3]
    N, sigma = KF.correct_key_case(ks)
    assert F.ints(sigma[:11]) == pyref.correct_key_prove(ks[0].p, ks[0].q)
    sigma[11 + 4, 0] ^= 1                                               # key 1: one of the 11 roots is wrong
    ok = np.zeros(3, dtype=np.uint8)
    orc.lib.orc_correct_key_verify(3, orc._p(N), orc._p(sigma), orc._p(ok))
    want = [pyref.correct_key_verify(F.ints(N[i:i + 1])[0], F.ints(sigma[11 * i:11 * i + 11])) for i in range(3)]
    assert list(ok) == [int(w) for w in want] == [1, 0, 1]
    # a modulus with a small prime factor is refused whatever the roots are
    bad = F.words([ks[0].p * 6361], 64)
    orc.lib.orc_correct_key_verify(1, orc._p(bad), orc._p(sigma[:11]), orc._p(ok))
    assert ok[0] == 0 and not pyref.correct_key_verify(ks[0].p * 6361, F.ints(sigma[:11]))


def test_composite_dlog_proof(keys):
    ks = keys[:4]
    N, g, ni, x, y = KF.composite_dlog_case(ks)
    y[1, 0] ^= 1
    x[2, 3] ^= 4
    ok = np.zeros(4, dtype=np.uint8)
    orc.lib.orc_composite_dlog_verify(4, *[orc._p(a) for a in (N, g, ni, x, y, ok)])
    want = [pyref.composite_dlog_verify(*[F.ints(a[i:i + 1])[0] for a in (N, g, ni, x, y)]) for i in range(4)]
    assert list(ok) == [int(w) for w in want] == [1, 0, 0, 1]


def test_feldman_shares():
    t, n, B = 2, 5, 3
    commits, shares, index, pts = KF.vss_case(t, n, B)
    shares[4, 0] ^= 1
    ok = np.zeros(B * n, dtype=np.uint8)
    orc.lib.orc_vss_validate_share(B * n, t + 1, orc._p(commits), orc._p(shares), orc._p(index), orc._p(ok))
    assert list(ok) == [0 if i == 4 else 1 for i in range(B * n)]
    out = orc.u32((B * n, 16))
    orc.lib.orc_vss_point_commitment(B * n, t + 1, orc._p(commits), orc._p(index), orc._p(out))
    assert F.points(out) == [pyref.vss_point(pts[i // n], int(index[i])) for i in range(B * n)]


def test_round1_verdict_as_the_reference_composes_it(keys):
    """party_i.rs:260-320 on the oracle: every conjunct refuses on its own, `bad_actors` names exactly the provers that failed, and the
    reference's test_small_paillier (gg_2020/test.rs:764-783) holds: a 2046-bit Paillier key with a VALID correct-key proof is refused"""
    n = 3
    c = KF.round1_case(keys, n, 4)
    ok, bad = KF.oracle_round1(c, n)
    assert ok.all() and not bad.any()
    p, q, Nsmall, sig_small = KF.small_paillier_key()
    assert F.ints(Nsmall)[0].bit_length() == 2046
    v = np.zeros(1, dtype=np.uint8)
    orc.lib.orc_correct_key_verify(1, orc._p(Nsmall), orc._p(sig_small), orc._p(v))
    assert v[0] == 1                                                   # the proof itself is fine: only the length check stands in the way
    t = {f: a.copy() for f, a in c.items()}
    t["com"][0, 3] ^= 1                                                # session 0, prover 0: wrong commitment
    t["blind"][4, 0] ^= 1                                              # session 1, prover 1: wrong decommitment
    t["N"][5], t["sigma"][5] = Nsmall[0], sig_small[0]                 # session 1, prover 2: small Paillier modulus
    t["sigma"][6, 70] ^= 1                                             # session 2, prover 0: bad NiCorrectKeyProof
    t["y_h1"][7, 2] ^= 1                                               # session 2, prover 1: bad proof for base h1
    t["x_h2"][8, 9] ^= 1                                               # session 2, prover 2: bad proof for base h2
    t["Nt"][9, 63] &= 0x3fffffff                                       # session 3, prover 0: N~ of 2046 bits (its proofs fail too)
    ok, bad = KF.oracle_round1(t, n)
    assert list(ok) == [0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1] and list(bad) == [0b001, 0b110, 0b111, 0b001]
    # test_small_paillier literally: one party, share_count 1
    one = {f: a[5:6].copy() for f, a in t.items()}
    ok1, bad1 = KF.oracle_round1(one, 1)
    assert list(ok1) == [0] and list(bad1) == [1]


def test_round2_verdict_as_the_reference_composes_it():
    """party_i.rs:322-367: validate_share && commitments[0] == y_vec[i]"""
    t, n, B = 1, 3, 4
    commits, shares, index, cm = KF.vss_case(t, n, B, seed="vss-r2")
    y = np.ascontiguousarray(commits[:, :16]).copy()
    shares[2, 0] ^= 1
    y[7, 1] ^= 1
    ok, bad = np.zeros(B * n, dtype=np.uint8), np.zeros(B, dtype=np.uint32)
    orc.lib.orc_keygen_verify_round2(B * n, n, t + 1, orc._p(commits), orc._p(shares), orc._p(index), orc._p(y), orc._p(ok), orc._p(bad))
    assert list(ok) == [0 if i in (2, 7) else 1 for i in range(B * n)] and list(bad) == [0b100, 0, 0b010, 0]
