"""The CPU oracle pinned against OpenSSL 1.1.1l (libcrypto — code nobody here wrote) where the image offers it: secp256k1
point arithmetic, ECDSA verification of the oracle's GG20 signatures under the wallet's public key (the reference's own
independent check is libsecp256k1 in gg_2020/test.rs:711-748 `check_sig`; SURVEY.md 8d configs 1 and 4 say "verifies under
OpenSSL"), SHA-256 known answers, and GMP's mpz_powm against BN_mod_exp.  What this does NOT pin: curv's byte encodings
inside the sigma-proof transcripts."""
import hashlib

import numpy as np

import fixtures as F
import gg20_fixture as G
import orc
import ossl
import pyref

Q = pyref.Q


def test_openssl_is_the_third_party_library():
    assert ossl.version().startswith("OpenSSL ")


def test_sha256_known_answers():
    # FIPS 180-2 appendix B vectors
    kat = {b"abc": "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad",
           b"": "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855",
           b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq": "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"}
    for m, h in kat.items():
        assert ossl.sha256(m).hex() == h == orc.sha256(m).hex() == hashlib.sha256(m).hexdigest()
    r = F.Rng("sha-kat")
    for ln in (1, 55, 56, 63, 64, 65, 119, 120, 1000):
        m = bytes(r.bits(8) for _ in range(ln))
        assert ossl.sha256(m) == orc.sha256(m)


def test_oracle_point_arithmetic_equals_openssl():
    r = F.Rng("ossl-ec")
    ks = [1, 2, 3, Q - 1, Q - 2, (Q + 1) // 2, 2**255 % Q, 2**128, 2**128 - 1] + [r.below(Q - 1) + 1 for _ in range(119)]
    k = F.words(ks, 8)
    P = ossl.ec_mul(k)
    assert np.array_equal(P, orc.ec_mul_base(k))
    # the generator itself, published coordinates (SEC 2, 2.4.1)
    assert F.ints(P[:1, :8])[0] == 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
    assert F.ints(P[:1, 8:])[0] == 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
    x = F.words([r.below(Q - 1) + 1 for _ in range(len(ks))], 8)
    assert np.array_equal(ossl.ec_mul(x, P), orc.ec_mul(x, P))
    Qp = np.roll(P, 1, axis=0)
    Qp[0] = P[0]                                   # a doubling
    neg = P[1].copy()
    neg[8:] = F.words([pyref.P - F.ints(P[1:2, 8:])[0]], 8)[0]
    Qp[1] = neg                                    # P + (-P) = infinity (all-zero words in both)
    assert np.array_equal(ossl.ec_add(P, Qp), orc.ec_add(P, Qp))
    assert not ossl.ec_add(P, Qp)[1].any()


def test_gmp_powm_equals_bn_mod_exp():
    r = F.Rng("ossl-powm")
    for bits in (2048, 4096):
        mods = [r.bits(bits) | (1 << (bits - 1)) | 1 for _ in range(3)]
        base = [r.bits(bits) for _ in range(6)]
        exp = [r.bits(bits // 2) for _ in range(6)]
        got = F.ints(orc.modexp(F.words(mods, bits // 32), F.words(base, bits // 32), F.words(exp, bits // 64), [i % 3 for i in range(6)]))
        assert got == [ossl.modexp(b % mods[i % 3], e, mods[i % 3]) for i, (b, e) in enumerate(zip(base, exp))]


def test_config1_oracle_signature_verifies_under_openssl(keys):
    """BASELINE config 1 / SURVEY.md 8d(1): one t=1, n=3 key, signers [1, 2], message = SHA-256("ZenGo") as in
    state_machine/sign.rs:744; the CPU path signs, OpenSSL verifies under y; any change to r, s or the message is rejected"""
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    nonces = G.make_nonces(lk, 1, seed="config1")
    msg = int.from_bytes(hashlib.sha256(b"ZenGo").digest(), "big")
    nonces["msg"] = F.words([msg], 8)
    r, s, recid, R, status = G.oracle_sign(lk, nonces, 1)
    assert list(status) == [0]
    y = lk["arrays"]["y"][0]
    assert ossl.ecdsa_verify(y, nonces["msg"], r, s).all()
    for arr in (r, s, nonces["msg"]):
        bad = arr.copy()
        bad[0, 0] ^= 1
        args = [nonces["msg"], r, s]
        args[[id(a) for a in (nonces["msg"], r, s)].index(id(arr))] = bad
        assert not ossl.ecdsa_verify(y, *args).any()
    other = G.make_local_keys(keys, 1, 3, [0, 1], seed="another wallet")["arrays"]["y"][0]
    assert not ossl.ecdsa_verify(other, nonces["msg"], r, s).any()
    # R.x mod q = r: the recovered nonce point of the protocol is the one the signature commits to
    assert F.ints(R[:, :8])[0] % Q == F.ints(r)[0]


def test_oracle_signatures_of_every_shape_verify_under_openssl(keys):
    for t, n, signers, B in [(1, 3, [0, 2], 6), (2, 5, [0, 2, 4], 2), (1, 3, [0, 1, 2], 2)]:
        lk = G.make_local_keys(keys, t, n, signers)
        nonces = G.make_nonces(lk, B, seed=f"ossl-{t}-{n}")
        r, s, recid, R, status = G.oracle_sign(lk, nonces, B)
        assert not status.any()
        assert ossl.ecdsa_verify(lk["arrays"]["y"][0], nonces["msg"], r, s).all()
        # low-s normalisation (party_i.rs:895-905): s <= q/2
        assert all(v <= Q // 2 for v in F.ints(s))
