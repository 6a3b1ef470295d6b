"""CPU test of the GG20 signing oracle (oracle/gg20_oracle.c): complete sessions for the reference's own
(t, n, signer-set) cases (gg_2020/state_machine/sign.rs:740-763; gg_2020/test.rs:55-67) must yield signatures
that verify under an independent ECDSA verifier (the reference uses libsecp256k1's, test.rs:711-748; here the
pure-Python one) — plus recid / low-s conventions of party_i.rs:873-910."""
import pytest

import fixtures as F
import gg20_fixture as G
import pyref


@pytest.mark.parametrize("t,n,signers", [(1, 3, [0, 1]), (1, 3, [0, 2]), (1, 3, [1, 2]), (2, 5, [0, 2, 4])])
def test_sign_verifies_independently(keys, t, n, signers):
    lk = G.make_local_keys(keys, t, n, signers)
    B = 2
    nonces = G.make_nonces(lk, B, seed=f"cpu-{t}-{n}-{signers}")
    r, s, recid, R, status = G.oracle_sign(lk, nonces, B)
    assert list(status) == [0] * B
    for b in range(B):
        rr, ss, m = F.ints(r[b:b + 1])[0], F.ints(s[b:b + 1])[0], F.ints(nonces["msg"][b:b + 1])[0]
        assert pyref.ecdsa_verify(lk["y"], m, rr, ss)
        assert ss <= pyref.Q // 2                                   # low-s normalisation
        Rp = F.points(R[b:b + 1])[0]
        assert rr == Rp[0] % pyref.Q
        # recovery id: parity of R.y, flipped when s was negated; recover the key and compare
        y_par = recid[b] & 1
        x = rr
        yy = pow(x ** 3 + 7, (pyref.P + 1) // 4, pyref.P)
        if yy & 1 != y_par:
            yy = pyref.P - yy
        rinv = pow(rr, -1, pyref.Q)
        rec = pyref.ec_mul(rinv, pyref.ec_add(pyref.ec_mul(ss, (x, yy)), pyref.ec_neg(pyref.ec_mul(m, pyref.G))))
        assert rec == lk["y"]


def test_inconsistent_public_key_is_detected(keys):
    """phase6_check_S_i_sum (party_i.rs:835-848): sum S_i must equal the group public key"""
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    nonces = G.make_nonces(lk, 1, seed="cpu-bad")
    lk["arrays"]["y"][:] = F.point_words([pyref.ec_mul(12345, pyref.G)])
    *_, status = G.oracle_sign(lk, nonces, 1)
    assert status[0] == 602
