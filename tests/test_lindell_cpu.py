"""Oracle self-test for the Lindell'17 signing path: the signatures the restatement produces verify under the
independent Python ECDSA check against the joint public key x1 x2 G (the reference's own test asserts
party_one::verify on the result, lindell_2017/test.rs), are low-s, and the recovery id recovers that key."""
import fixtures as F
import lindell_fixture as L
import pyref


def test_lindell_signatures_verify(keys):
    B = 12
    fx = L.make(keys, B)
    c3, r, s, recid = L.oracle_run(fx)
    for i in range(B):
        ri, si = F.ints(r[i:i + 1])[0], F.ints(s[i:i + 1])[0]
        assert 0 < si <= pyref.Q // 2
        assert pyref.ecdsa_verify(fx["pub"][i], fx["msg_int"][i] % pyref.Q, ri, si)
        assert int(recid[i]) in (0, 1)


def test_oracle_matches_the_independent_python_restatement(keys):
    """c3 and (r, s, recid) of the C/GMP oracle equal the pure-Python restatement (different big-integer engine,
    textbook instead of CRT decryption) on the same inputs, including the edge messages of the fixture."""
    B = 8
    fx = L.make(keys, B, seed="lindell-pyref")
    c3, r, s, recid = L.oracle_run(fx)
    ints = lambda a: F.ints(a)
    pts = F.points(fx["R1"]), F.points(fx["R2"])
    for i in range(B):
        k = keys[fx["kidx"][i]]
        want_c3 = pyref.lindell_partial_sig(k.N, ints(fx["c_key"])[i], ints(fx["x2"])[i], ints(fx["k2"])[i], pts[0][i],
                                            ints(fx["msg"])[i], ints(fx["rho"])[i], ints(fx["r"])[i])
        assert ints(c3)[i] == want_c3
        assert (ints(r)[i], ints(s)[i], int(recid[i])) == pyref.lindell_sign(k.p, k.q, want_c3, ints(fx["k1"])[i], pts[1][i])
