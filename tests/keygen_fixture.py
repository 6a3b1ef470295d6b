"""Fixtures for the keygen verification math: proofs a party would broadcast in keygen (NiCorrectKeyProof, the two
CompositeDLogProofs for h1 / h2, Feldman VSS) built by the ORACLE's prove side from the key material of tests/golden/keys16.json."""
import numpy as np

import fixtures as F
import orc
import pyref


def correct_key_case(keys):
    """(N [K][64], sigma [K][11][64]) for the Paillier keys of the fixtures"""
    K = len(keys)
    sigma = orc.u32((K * 11, 64))
    orc.lib.orc_correct_key_prove(K, orc._p(F.words([k.p for k in keys], 32)), orc._p(F.words([k.q for k in keys], 32)), orc._p(sigma))
    return F.words([k.N for k in keys], 64), sigma


def composite_dlog_case(keys, seed="cdlog"):
    """statements (N~, g, ni = g^-secret) with their proofs: the fixtures' h2 is h1^xi for an unknown xi, so the case mints
    its own ni from a fresh secret (generate_h1_h2_N_tilde, party_i.rs:137-156: h2 = h1^xhi, the proof is for -xhi mod phi)"""
    r = F.Rng(seed)
    N, g, sec, nonce = [], [], [], []
    for k in keys:
        N.append(k.Nt)
        g.append(k.h1)
        sec.append(r.below(k.Nt >> 2))
        nonce.append(r.bits(512))
    ni = [pow(pow(gg, s, n), -1, n) for gg, s, n in zip(g, sec, N)]
    Nw, gw, nw = F.words(N, 64), F.words(g, 64), F.words(ni, 64)
    x, y = orc.u32((len(keys), 64)), orc.u32((len(keys), 73))
    orc.lib.orc_composite_dlog_prove(len(keys), orc._p(Nw), orc._p(gw), orc._p(nw), orc._p(F.words(sec, 64)), orc._p(F.words(nonce, 16)), orc._p(x), orc._p(y))
    return Nw, gw, nw, x, y


def vss_case(t, n, B, seed="vss"):
    """B sharings of degree t among n parties: commitments [B][t+1][16], for every (sharing, party) the share and index"""
    r = F.Rng(seed)
    commits, shares, index = [], [], []
    for b in range(B):
        coef = [r.below(pyref.Q - 1) + 1 for _ in range(t + 1)]
        commits.append([pyref.ec_mul(c, pyref.G) for c in coef])
        for i in range(1, n + 1):
            shares.append(sum(c * pow(i, e, pyref.Q) for e, c in enumerate(coef)) % pyref.Q)
            index.append(i)
    cw = np.concatenate([F.point_words(c) for c in commits for _ in range(n)])          # one row of commitments per (sharing, party)
    return cw.reshape(B * n, (t + 1) * 16), F.words(shares, 8), np.array(index, dtype=np.int32), commits



def round1_case(keys, n_parties, sessions, seed="kg-r1"):
    """What `sessions` keygen sessions of `n_parties` parties broadcast in round 1 (KeyGenBroadcastMessage1 + KeyGenDecommitMessage1,
    party_i.rs:219-258) — built by the oracle's PROVE side from the fixture keys exactly as the reference does it: the secrets of the
    two CompositeDLogProofs are phi - xhi and phi - xhi^-1 mod phi (generate_h1_h2_N_tilde, party_i.rs:137-156).  Returns a dict of
    uint32 arrays, items = (session, prover)."""
    import json
    import os
    with open(os.path.join(F.HERE, "golden", "keys16.json")) as f:
        raw = json.load(f)["keys"]
    r = F.Rng(seed)
    B = n_parties * sessions
    ix = [i % len(keys) for i in range(B)]
    ks = [keys[i] for i in ix]
    N, sigma = correct_key_case(ks)
    y, blind, sec1, sec2, n1, n2 = [], [], [], [], [], []
    for i in ix:
        P, Q, xhi = int(raw[i]["nt_p"], 16), int(raw[i]["nt_q"], 16), int(raw[i]["xhi"], 16)
        phi = (P - 1) * (Q - 1)
        y.append(pyref.ec_mul(r.below(pyref.Q - 1) + 1, pyref.G))
        blind.append(r.bits(256))
        sec1.append(phi - xhi)
        sec2.append(phi - pow(xhi, -1, phi))
        n1.append(r.bits(512)); n2.append(r.bits(512))
    yw, bw = F.point_words(y), F.words(blind, 8)
    com = orc.u32((B, 8))
    orc.lib.orc_hash_commit_point(B, orc._p(yw), orc._p(bw), orc._p(com))
    Ntw, h1w, h2w = F.words([k.Nt for k in ks], 64), F.words([k.h1 for k in ks], 64), F.words([k.h2 for k in ks], 64)
    x1, y1, x2, y2 = orc.u32((B, 64)), orc.u32((B, 73)), orc.u32((B, 64)), orc.u32((B, 73))
    orc.lib.orc_composite_dlog_prove(B, orc._p(Ntw), orc._p(h1w), orc._p(h2w), orc._p(F.words(sec1, 64)), orc._p(F.words(n1, 16)), orc._p(x1), orc._p(y1))
    orc.lib.orc_composite_dlog_prove(B, orc._p(Ntw), orc._p(h2w), orc._p(h1w), orc._p(F.words(sec2, 64)), orc._p(F.words(n2, 16)), orc._p(x2), orc._p(y2))
    return dict(y=yw, blind=bw, com=com, N=N, sigma=sigma.reshape(B, 11 * 64), Nt=Ntw, h1=h1w, h2=h2w, x_h1=x1, y_h1=y1, x_h2=x2, y_h2=y2)


def oracle_round1(case, n_parties):
    B = case["N"].shape[0]
    ok, bad = np.zeros(B, dtype=np.uint8), np.zeros(B // n_parties, dtype=np.uint32)
    orc.lib.orc_keygen_verify_round1(B, n_parties, *[orc._p(np.ascontiguousarray(case[f])) for f in
                                     ("y", "blind", "com", "N", "sigma", "Nt", "h1", "h2", "x_h1", "y_h1", "x_h2", "y_h2")], orc._p(ok), orc._p(bad))
    return ok, bad


def small_paillier_key(bits=2046, seed="small-paillier"):
    """a Paillier key of `bits` bits with its NiCorrectKeyProof — the reference's test_small_paillier (gg_2020/test.rs:764-783)
    builds one with keypair_with_modulus_size(2046): two 1023-bit primes whose product has exactly 2046 bits"""
    r = F.Rng(seed)
    half = bits // 2
    while True:
        ps = []
        for _ in range(2):
            start = (r.bits(half - 2) | (1 << (half - 2))) | (1 << (half - 1))          # top two bits set: the product has 2 * half bits
            out = orc.u32((1, 32))
            orc.lib.orc_nextprime(32, orc._p(F.words([start], 32)), orc._p(out))
            ps.append(F.ints(out)[0])
        if ps[0] != ps[1] and (ps[0] * ps[1]).bit_length() == bits:
            break
    sigma = orc.u32((11, 64))
    orc.lib.orc_correct_key_prove(1, orc._p(F.words([ps[0]], 32)), orc._p(F.words([ps[1]], 32)), orc._p(sigma))
    return ps[0], ps[1], F.words([ps[0] * ps[1]], 64), sigma.reshape(1, 11 * 64)
