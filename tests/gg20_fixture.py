"""GG20 signing fixtures: a `LocalKey` set (Shamir shares of the ECDSA key + the Paillier / N~ material of
tests/golden/keys16.json, as the reference's keygen would leave it: keygen/rounds.rs:283-322) and per-session
nonces drawn with the reference's ranges from a seeded stream.  Array layouts = oracle/mpe_oracle.h."""
import ctypes as C

import numpy as np

import fixtures as F
import pyref

Q = pyref.Q


class KeysStruct(C.Structure):
    _fields_ = [("t", C.c_int), ("n", C.c_int), ("S", C.c_int), ("signers", C.c_void_p)] + \
               [(f, C.c_void_p) for f in ("x", "p", "q", "Nt", "h1", "h2", "y", "X")]


NONCE_FIELDS = ["k", "gamma", "blind", "r_a", "al_alpha", "al_beta", "al_gamma", "al_rho", "mb_beta_tag", "mb_r",
                "mb_nonce_b", "mb_nonce_bt", "l", "ped_s1", "ped_s2", "pdl_alpha", "pdl_beta", "pdl_rho", "pdl_gamma",
                "heg_s1", "heg_s2", "msg"]


class NoncesStruct(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in NONCE_FIELDS]


def make_local_keys(keys, t, n, signers, seed="keygen"):
    """Shamir-share a secret x with a degree-t polynomial (the sum polynomial of GG20 keygen); party i holds
    x_i = F(i+1), pk_vec[i] = x_i G, y = x G; Paillier/N~ material of party i = keys[i]."""
    r = F.Rng(seed)
    coef = [r.below(Q - 1) + 1 for _ in range(t + 1)]
    xs = [sum(c * pow(i + 1, e, Q) for e, c in enumerate(coef)) % Q for i in range(n)]
    y = pyref.ec_mul(coef[0], pyref.G)
    X = [pyref.ec_mul(x, pyref.G) for x in xs]
    arr = dict(signers=np.array(signers, dtype=np.int32), x=F.words(xs, 8), p=F.words([keys[i].p for i in range(n)], 32),
               q=F.words([keys[i].q for i in range(n)], 32), Nt=F.words([keys[i].Nt for i in range(n)], 64),
               h1=F.words([keys[i].h1 for i in range(n)], 64), h2=F.words([keys[i].h2 for i in range(n)], 64),
               y=F.point_words([y]), X=F.point_words(X))
    return dict(t=t, n=n, S=len(signers), arrays=arr, y=y, x=coef[0], keys=keys[:n])


def keys_struct(lk):
    s = KeysStruct()
    s.t, s.n, s.S = lk["t"], lk["n"], lk["S"]
    for f, a in lk["arrays"].items():
        setattr(s, f, a.ctypes.data)
    return s


def make_nonces(lk, B, seed="sessions"):
    """Every value the reference samples during signing (party_i.rs:559-563,574,627; mta/mod.rs:57,97-98;
    range_proofs.rs:48-51; zk_pdl_with_slack/mod.rs:73-77; curv sigma proofs), per session."""
    r = F.Rng(seed)
    S, n, keys = lk["S"], lk["n"], lk["keys"]
    signers = list(lk["arrays"]["signers"])
    P = S * (S - 1)
    q3 = Q ** 3
    z = {f: [] for f in NONCE_FIELDS}
    sc = lambda: r.below(Q - 1) + 1
    for b in range(B):
        for i in range(S):
            me = keys[signers[i]]
            z["k"].append(sc()); z["gamma"].append(sc()); z["blind"].append(r.bits(256)); z["r_a"].append(r.below(me.N))
            z["l"].append(sc()); z["ped_s1"].append(sc()); z["ped_s2"].append(sc()); z["heg_s1"].append(sc()); z["heg_s2"].append(sc())
            for st in range(n):
                nn = F.alice_nonces(r, me, keys[st])
                for f in ("alpha", "beta", "gamma", "rho"):
                    z["al_" + f].append(nn[f])
            for jj in range(S - 1):
                ind = jj if jj < i else jj + 1
                alice = keys[signers[ind]]
                for v in range(2):
                    z["mb_beta_tag"].append(r.below(alice.N)); z["mb_r"].append(r.below(alice.N))
                    z["mb_nonce_b"].append(sc()); z["mb_nonce_bt"].append(sc())
                nn = F.pdl_nonces(r, me, keys[signers[ind]])
                for f in ("alpha", "beta", "rho", "gamma"):
                    z["pdl_" + f].append(nn[f])
        z["msg"].append(int.from_bytes(__import__("hashlib").sha256(b"session %d" % b).digest(), "big"))
    widths = dict(k=8, gamma=8, blind=8, r_a=64, al_alpha=24, al_beta=64, al_gamma=88, al_rho=72, mb_beta_tag=64, mb_r=64,
                  mb_nonce_b=8, mb_nonce_bt=8, l=8, ped_s1=8, ped_s2=8, pdl_alpha=24, pdl_beta=64, pdl_rho=72, pdl_gamma=88,
                  heg_s1=8, heg_s2=8, msg=8)
    return {f: F.words(v, widths[f]) for f, v in z.items()}


def nonces_struct(arrs):
    s = NoncesStruct()
    for f in NONCE_FIELDS:
        setattr(s, f, arrs[f].ctypes.data)
    return s


def oracle_sign(lk, nonces, B, first=0, count=None):
    import orc
    count = B if count is None else count
    r, s = np.zeros((B, 8), dtype=np.uint32), np.zeros((B, 8), dtype=np.uint32)
    recid, status = np.zeros(B, dtype=np.int32), np.full(B, -1, dtype=np.int32)
    R = np.zeros((B, 16), dtype=np.uint32)
    ks, ns = keys_struct(lk), nonces_struct(nonces)
    orc.lib.orc_gg20_sign(C.byref(ks), C.byref(ns), first, count, orc._p(r), orc._p(s), orc._p(recid), orc._p(R), orc._p(status))
    return r, s, recid, R, status
