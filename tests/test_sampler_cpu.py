"""The sampler's CPU restatement (oracle/sampler_oracle.c) against an independent pure-Python restatement of the same rules —
curv `Samplable` (sample, sample_below, sample_range), the reference's `from_modulo` (src/utilities/mta/range_proofs.rs:538-557),
`Scalar::random` — over ChaCha20 (RFC 8439 block function, pinned by the RFC's own test vector), and the distributions the
signing path needs (range_proofs.rs:48-51; zk_pdl_with_slack/mod.rs:73-77; mta/mod.rs:57,97-98; party_i.rs:561-563,574,628)."""
import hashlib
import math
import struct

import numpy as np

import fixtures as F
import gg20_fixture as G
import orc

Q = F.Q


# ---- an independent ChaCha20 + curv Samplable in Python --------------------------------------------------------------
def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & 0xffffffff


def _qr(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & 0xffffffff; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & 0xffffffff; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & 0xffffffff; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & 0xffffffff; s[b] = _rotl(s[b] ^ s[c], 7)


def py_chacha_block(key, counter, n13, n14, n15):
    init = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(struct.unpack("<8I", key)) + [counter, n13, n14, n15]
    s = list(init)
    for _ in range(10):
        _qr(s, 0, 4, 8, 12); _qr(s, 1, 5, 9, 13); _qr(s, 2, 6, 10, 14); _qr(s, 3, 7, 11, 15)
        _qr(s, 0, 5, 10, 15); _qr(s, 1, 6, 11, 12); _qr(s, 2, 7, 8, 13); _qr(s, 3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & 0xffffffff for a, b in zip(s, init)])


class PyStream:
    """fill_bytes of item `item` of stream `sid`"""

    def __init__(self, seed, item, sid):
        self.seed, self.item, self.sid, self.pos, self.buf, self.draws = seed, item, sid, 0, b"", 0

    def fill(self, n):
        while len(self.buf) < self.pos + n:
            self.buf += py_chacha_block(self.seed, len(self.buf) // 64, self.item, self.sid & 0xffffffff, self.sid >> 32)
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out

    def sample(self, bits):                                   # BigInt::sample
        nbytes = (bits - 1) // 8 + 1
        self.draws += 1
        return int.from_bytes(self.fill(nbytes), "big") >> (nbytes * 8 - bits)

    def below(self, upper, nonzero=False, coprime=False):     # sample_below (+ the call sites' extra conditions)
        while True:
            x = self.sample(upper.bit_length())
            if x < upper and not (nonzero and x == 0) and not (coprime and math.gcd(x, upper) != 1):
                return x


SEED = bytes(range(32))


def test_chacha20_block_is_rfc8439():
    # RFC 8439 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00
    want = bytes.fromhex("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                         "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    assert orc.chacha20_block(SEED, 1, 0x09000000, 0x4a000000, 0) == want
    assert py_chacha_block(SEED, 1, 0x09000000, 0x4a000000, 0) == want


def test_sample_bits_is_curv_sample():
    for bits, words in ((256, 8), (1, 1), (7, 1), (33, 2), (2047, 64), (2816, 88)):
        got = F.ints(orc.sample_bits(5, SEED, 0x1234, bits, words))
        want = [PyStream(SEED, i, 0x1234).sample(bits) for i in range(5)]
        assert got == want and all(v < (1 << bits) for v in got)


def test_sample_below_repeats_until_below_and_counts_every_draw():
    keys = F.load_keys()
    # bounds with very different rejection rates: 2^k + 1 rejects ~half of the draws, 2^k - 1 next to none
    bounds = [keys[0].N, (1 << 2047) + 1, (1 << 300) - 1, Q ** 3, Q * keys[1].Nt, Q ** 3 * keys[2].Nt, 3, 2, 1]
    for u in bounds:
        words = (u.bit_length() + 31) // 32
        got, fails = orc.sample_below(64, SEED, 7, F.words([u], words), words)
        streams = [PyStream(SEED, i, 7) for i in range(64)]
        want = [s.below(u) for s in streams]
        assert fails == 0 and F.ints(got) == want and all(v < u for v in want)
        if u == (1 << 2047) + 1:
            assert sum(s.draws for s in streams) > 64 * 1.5        # the rejected draws were really taken from the stream
    # per-item bounds through an index
    tab = F.words([keys[i].N for i in range(4)], 64)
    idx = [i % 4 for i in range(32)]
    got, _ = orc.sample_below(32, SEED, 8, tab, 64, bound_idx=idx)
    assert F.ints(got) == [PyStream(SEED, i, 8).below(keys[i % 4].N) for i in range(32)]


def test_flags_nonzero_plus_one_coprime():
    # Scalar::random(): 0 < x < q
    got, fails = orc.sample_scalar(200, SEED, 9)
    assert fails == 0 and F.ints(got) == [PyStream(SEED, i, 9).below(Q, nonzero=True) for i in range(200)]
    assert all(0 < v < Q for v in F.ints(got))
    # a tiny bound makes the zero rejection visible
    got, _ = orc.sample_below(64, SEED, 10, F.words([3], 1), 1, flags=orc.SAMPLE_NONZERO)
    assert F.ints(got) == [PyStream(SEED, i, 10).below(3, nonzero=True) for i in range(64)] and set(F.ints(got)) == {1, 2}
    # sample_range(1, N - 1) = 1 + sample_below(N - 2)   (zk_pdl_with_slack/mod.rs:75)
    N = F.load_keys()[0].N
    got, _ = orc.sample_below(16, SEED, 11, F.words([N - 2], 64), 64, flags=orc.SAMPLE_PLUS_ONE)
    assert F.ints(got) == [1 + PyStream(SEED, i, 11).below(N - 2) for i in range(16)]
    # from_modulo on a modulus full of small factors: most candidates are refused by the gcd
    smooth = 1
    for p in (3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47):
        smooth *= p
    smooth = smooth ** 32
    assert smooth.bit_length() <= 2048
    got, fails = orc.sample_below(64, SEED, 12, F.words([smooth], 64), 64, flags=orc.SAMPLE_COPRIME)
    streams = [PyStream(SEED, i, 12) for i in range(64)]
    want = [s.below(smooth, coprime=True) for s in streams]
    assert fails == 0 and F.ints(got) == want and all(math.gcd(v, smooth) == 1 for v in want)
    assert sum(s.draws for s in streams) > 64 * 3
    # an even "modulus" can never be a Paillier key: every item fails, rows are zero
    got, fails = orc.sample_below(4, SEED, 13, F.words([1 << 100], 4), 4, flags=orc.SAMPLE_COPRIME)
    assert fails == 4 and not got.any()


def test_gg20_nonces_follow_the_references_ranges():
    keys = F.load_keys()
    for (t, n, signers, local) in ((1, 3, [0, 2], None), (2, 5, [0, 2, 4], [1])):
        lk = G.make_local_keys(keys, t, n, signers)
        B, S = 3, len(signers)
        loc = list(range(S)) if local is None else local
        z, fails = G.oracle_sample_nonces(lk, B, SEED, 5, local=local)
        assert fails == 0
        v = {f: F.ints(a) for f, a in z.items()}
        for f in ("k", "gamma", "l", "ped_s1", "ped_s2", "heg_s1", "heg_s2", "mb_nonce_b", "mb_nonce_bt"):
            assert all(0 < x < Q for x in v[f]), f
        assert all(x < (1 << 256) for x in v["blind"])
        L, P1 = len(loc), S - 1
        for b in range(B):
            for li, i in enumerate(loc):
                pi = b * L + li
                me = keys[signers[i]]
                assert v["r_a"][pi] < me.N
                for st in range(n):
                    ap = pi * n + st
                    assert v["al_alpha"][ap] < Q ** 3 and v["al_beta"][ap] < me.N and math.gcd(v["al_beta"][ap], me.N) == 1
                    assert v["al_gamma"][ap] < Q ** 3 * keys[st].Nt and v["al_rho"][ap] < Q * keys[st].Nt
                for jj in range(P1):
                    pp = pi * P1 + jj
                    peer = keys[signers[jj if jj < i else jj + 1]]
                    for w in range(2):
                        assert v["mb_beta_tag"][pp * 2 + w] < peer.N and v["mb_r"][pp * 2 + w] < peer.N
                    assert v["pdl_alpha"][pp] < Q ** 3 and 1 <= v["pdl_beta"][pp] <= me.N - 2
                    assert v["pdl_rho"][pp] < Q * peer.Nt and v["pdl_gamma"][pp] < Q ** 3 * peer.Nt
        # the composition is the primitives: field f of batch counter c is stream c | f << 56, item = the row's index in the layout
        # with EVERY signer local, (session * S + signer ordinal) * items per party + sub-item — the ordinal is part of the stream
        assert F.ints(z["k"]) == [PyStream(SEED, (pi // L) * S + loc[pi % L], 5 | (0 << 56)).below(Q, nonzero=True) for pi in range(B * L)]
        assert v["pdl_gamma"][-1] == PyStream(SEED, ((B - 1) * S + loc[-1]) * P1 + P1 - 1, 5 | (18 << 56)).below(Q ** 3 * keys[signers[(lambda i, jj: jj if jj < i else jj + 1)(loc[-1], P1 - 1)]].Nt)
        # deterministic in (seed, counter); another counter or seed gives other values everywhere
        z2, _ = G.oracle_sample_nonces(lk, B, SEED, 5, local=local)
        z3, _ = G.oracle_sample_nonces(lk, B, SEED, 6, local=local)
        z4, _ = G.oracle_sample_nonces(lk, B, bytes(32), 5, local=local)
        for f in G.NONCE_FIELDS[:-1]:
            assert np.array_equal(z[f], z2[f]) and not np.array_equal(z[f], z3[f]) and not np.array_equal(z[f], z4[f]), f


def test_objects_hosting_different_parties_never_share_a_stream():
    """two objects that hold different parties of one batch and get the SAME (seed, counter) draw different values, and each draws
    exactly what the all-local object draws for its party (the signer ordinal is part of the stream identity)"""
    keys = F.load_keys()
    t, n, signers = 2, 5, [0, 2, 4]
    lk = G.make_local_keys(keys, t, n, signers)
    B, S = 3, len(signers)
    P1 = S - 1
    full, _ = G.oracle_sample_nonces(lk, B, SEED, 9)
    per = {"k": 1, "gamma": 1, "blind": 1, "r_a": 1, "al_alpha": n, "al_beta": n, "al_gamma": n, "al_rho": n, "mb_beta_tag": 2 * P1, "mb_r": 2 * P1,
           "mb_nonce_b": 2 * P1, "mb_nonce_bt": 2 * P1, "l": 1, "ped_s1": 1, "ped_s2": 1, "pdl_alpha": P1, "pdl_beta": P1, "pdl_rho": P1,
           "pdl_gamma": P1, "heg_s1": 1, "heg_s2": 1}
    seen = {}
    for ordinal in range(S):
        one, _ = G.oracle_sample_nonces(lk, B, SEED, 9, local=[ordinal])
        for f, m in per.items():
            a = full[f].reshape(B, S, m, -1)[:, ordinal]
            assert np.array_equal(one[f].reshape(B, m, -1), a), (f, ordinal)
        seen[ordinal] = one["k"].tobytes()
    assert len(set(seen.values())) == S


def test_a_rejection_loop_that_gives_up_poisons_the_partys_k():
    """with one attempt per draw (curv would loop on: a deliberate divergence) many draws below N give up: the value is zero, the
    failure is counted, and the k_i of that (session, party) becomes an invalid scalar — the signing oracle answers status 91"""
    keys = F.load_keys()
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    B = 16
    msg = F.words([int.from_bytes(hashlib.sha256(b"give up %d" % b).digest(), "big") for b in range(B)], 8)
    try:
        orc.lib.orc_sampler_set_max_attempts(1)
        z, fails = G.oracle_sample_nonces(lk, B, SEED, 3, msg=msg)
    finally:
        orc.lib.orc_sampler_set_max_attempts(128)
    assert fails > 0
    k = F.ints(z["k"])
    poisoned = [pi for pi in range(B * 2) if k[pi] == (1 << 256) - 1]
    assert poisoned and all(0 < k[pi] < Q for pi in range(B * 2) if pi not in poisoned)
    r, s, recid, R, status = G.oracle_sign(lk, z, B)
    bad_sessions = sorted({pi // 2 for pi in poisoned})
    assert [b for b in range(B) if status[b] != 0] == bad_sessions and all(status[b] == 91 for b in bad_sessions)
    assert not r[bad_sessions].any() and not s[bad_sessions].any()


def test_sessions_signed_from_sampled_nonces_verify():
    """the sampled arrays feed the signing oracle unchanged: every session signs and the signature verifies (sign.rs:715-719)"""
    import hashlib
    import pyref
    keys = F.load_keys()
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    B = 2
    msg = F.words([int.from_bytes(hashlib.sha256(b"sampled %d" % b).digest(), "big") for b in range(B)], 8)
    z, fails = G.oracle_sample_nonces(lk, B, SEED, 77, msg=msg)
    assert fails == 0
    r, s, recid, R, status = G.oracle_sign(lk, z, B)
    assert list(status) == [0, 0]
    for b in range(B):
        assert pyref.ecdsa_verify(lk["y"], F.ints(msg[b:b + 1])[0] % Q, F.ints(r[b:b + 1])[0], F.ints(s[b:b + 1])[0])
