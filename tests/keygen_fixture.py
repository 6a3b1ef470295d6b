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
