"""Seeded inputs of the Lindell'17 signing path (lindell_2017/test.rs:test_two_party_sign shape): per item a key pair
x1, x2 (the public key is x1 x2 G), party one's Paillier key with c_key = Enc(x1), the ephemeral pairs k1, k2 and every
value the reference samples inside PartialSig::compute (rho < q^2, the encryption randomness)."""
import numpy as np

import fixtures as F
import orc
import pyref


def make(keys, B, seed="lindell"):
    r = F.Rng(seed)
    nk = len(keys)
    kidx = [(3 * i) % nk for i in range(B)]
    x1 = [r.below(pyref.Q // 3) + 1 for _ in range(B)]                 # party_one.rs:155 samples x1 below q/3
    x2 = [r.below(pyref.Q - 1) + 1 for _ in range(B)]
    k1 = [r.below(pyref.Q - 1) + 1 for _ in range(B)]
    k2 = [r.below(pyref.Q - 1) + 1 for _ in range(B)]
    msg = [r.bits(256) for _ in range(B)]
    rho = [r.below(pyref.Q ** 2) for _ in range(B)]
    rr = [r.below(keys[kidx[i]].N) for i in range(B)]
    r0 = [r.below(keys[kidx[i]].N) for i in range(B)]
    if B > 3:
        msg[1] = 0
        rho[2] = 0
        msg[3] = pyref.Q + 5                                            # a message above q is reduced by mod_mul
    N = F.words([k.N for k in keys], 64)
    c_key = orc.paillier_encrypt(N, F.words(x1, 64), F.words(r0, 64), kidx)
    k1w, k2w = F.words(k1, 8), F.words(k2, 8)
    R1, R2 = orc.ec_mul_base(k1w), orc.ec_mul_base(k2w)
    pub = [pyref.ec_mul(x1[i] * x2[i] % pyref.Q, pyref.G) for i in range(B)]
    return dict(kidx=kidx, N=N, p=F.words([k.p for k in keys], 32), q=F.words([k.q for k in keys], 32), c_key=c_key,
                x2=F.words(x2, 8), k1=k1w, k2=k2w, R1=R1, R2=R2, msg=F.words(msg, 8), rho=F.words(rho, 16),
                r=F.words(rr, 64), pub=pub, msg_int=msg)


def oracle_run(fx):
    c3 = orc.lindell_partial_sig(fx["N"], fx["c_key"], fx["x2"], fx["k2"], fx["R1"], fx["msg"], fx["rho"], fx["r"], fx["kidx"])
    r, s, recid = orc.lindell_sign(fx["p"], fx["q"], c3, fx["k1"], fx["R2"], fx["kidx"])
    return c3, r, s, recid
