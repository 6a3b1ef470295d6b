"""Shared test fixtures: deterministic key material (tests/golden/keys16.json) and seeded samplers
that follow the reference's nonce ranges (curv `BigInt::sample_below`)."""
import hashlib
import json
import os

import numpy as np

import pyref

HERE = os.path.dirname(os.path.abspath(__file__))
Q = pyref.Q


class Key:
    def __init__(self, d):
        self.p, self.q = int(d["p"], 16), int(d["q"], 16)
        self.N = self.p * self.q
        self.NN = self.N * self.N
        self.Nt, self.h1, self.h2 = int(d["n_tilde"], 16), int(d["h1"], 16), int(d["h2"], 16)


def load_keys():
    with open(os.path.join(HERE, "golden", "keys16.json")) as f:
        return [Key(d) for d in json.load(f)["keys"]]


class Rng:
    """SHA-256 counter stream; sample_below(u) = rejection sampling on bit_length(u) bits."""

    def __init__(self, seed):
        self.seed, self.ctr = str(seed).encode(), 0

    def bits(self, n):
        nbytes = (n + 7) // 8
        out = b""
        while len(out) < nbytes:
            out += hashlib.sha256(b"mpecdsa-test|" + self.seed + b"|" + self.ctr.to_bytes(8, "big")).digest()
            self.ctr += 1
        return int.from_bytes(out[:nbytes], "big") >> (nbytes * 8 - n)

    def below(self, u):
        n = u.bit_length()
        while True:
            x = self.bits(n)
            if x < u:
                return x

    def coprime_below(self, n):
        """`BigInt::from_modulo` (range_proofs.rs:544-552)"""
        import math
        while True:
            x = self.below(n)
            if math.gcd(x, n) == 1:
                return x


def words(vals, nwords):
    buf = b"".join(int(v).to_bytes(nwords * 4, "little") for v in vals)
    return np.frombuffer(buf, dtype="<u4").reshape(len(vals), nwords).copy()


def ints(arr):
    a = np.ascontiguousarray(arr, dtype="<u4")
    nb = a.shape[1] * 4
    raw = a.tobytes()
    return [int.from_bytes(raw[i * nb:(i + 1) * nb], "little") for i in range(a.shape[0])]


def point_words(pts):
    return words([(0 if p is None else p[0]) | ((0 if p is None else p[1]) << 256) for p in pts], 16)


def points(arr):
    out = []
    for v in ints(arr):
        x, y = v & ((1 << 256) - 1), v >> 256
        out.append(None if (x == 0 and y == 0) else (x, y))
    return out


def alice_nonces(rng, key_ek, key_st):
    q3 = Q ** 3
    return dict(alpha=rng.below(q3), beta=rng.coprime_below(key_ek.N), gamma=rng.below(q3 * key_st.Nt),
                rho=rng.below(Q * key_st.Nt))


def pdl_nonces(rng, key_ek, key_st):
    q3 = Q ** 3
    return dict(alpha=rng.below(q3), beta=1 + rng.below(key_ek.N - 2), rho=rng.below(Q * key_st.Nt),
                gamma=rng.below(q3 * key_st.Nt))


def bob_nonces(rng, key_ek, key_st):
    q3 = Q ** 3
    return dict(alpha=rng.below(q3), beta=rng.coprime_below(key_ek.N), gamma=rng.below(Q * Q * key_ek.N),
                rho=rng.below(Q * key_st.Nt), rho_prim=rng.below(q3 * key_st.Nt), sigma=rng.below(Q * key_st.Nt),
                tau=rng.below(q3 * key_st.Nt))
