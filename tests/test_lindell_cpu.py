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
