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
