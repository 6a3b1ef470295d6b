"""CPU tests: pin the C/GMP oracle (oracle/mpe_oracle.c) against the independent pure-Python
restatement (tests/pyref.py), published SHA-256 / secp256k1 vectors, and the reference's own
round-trip properties (range_proofs.rs:615-709, zk_pdl_with_slack/test.rs:11-129,
mta/test.rs:6-19)."""
import hashlib

import numpy as np

import fixtures as F
import orc
import pyref


def test_sha256_published_vectors():
    assert orc.sha256(b"abc").hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    assert orc.sha256(b"").hex() == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"
    m = bytes(range(256)) * 9
    for n in (55, 56, 63, 64, 65, 119, 120, 2304):
        assert orc.sha256(m[:n]) == hashlib.sha256(m[:n]).digest()


def test_secp256k1_known_multiples():
    # 2G and 3G (SEC2 / widely published)
    k = F.words([1, 2, 3, pyref.Q - 1, pyref.Q, pyref.Q + 5], 8)
    got = F.points(orc.ec_mul_base(k))
    assert got[0] == pyref.G
    assert got[1] == (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5,
                      0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A)
    assert got[2] == (0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9,
                      0x388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672)
    assert got[3] == pyref.ec_neg(pyref.G)
    assert got[4] is None                      # q*G = infinity (scalars reduce mod q)
    assert got[5] == pyref.ec_mul(5, pyref.G)
    assert pyref.H2[1] ** 2 % pyref.P == (pyref.H2[0] ** 3 + 7) % pyref.P


def test_ec_mul_add_vs_python():
    r = F.Rng("ec")
    ks = [r.below(pyref.Q) for _ in range(6)]
    ps = [pyref.ec_mul(r.below(pyref.Q), pyref.G) for _ in range(6)]
    got = F.points(orc.ec_mul(F.words(ks, 8), F.point_words(ps)))
    assert got == [pyref.ec_mul(k, p) for k, p in zip(ks, ps)]
    qs = ps[1:] + [pyref.ec_neg(ps[-1])]
    qs[0] = ps[0]                              # doubling case
    got = F.points(orc.ec_add(F.point_words(ps), F.point_words(qs)))
    assert got == [pyref.ec_add(a, b) for a, b in zip(ps, qs)]
    comp = orc.ec_compress(F.point_words(ps))
    assert [bytes(c) for c in comp] == [pyref.pt_bytes(p, True) for p in ps]


def test_modexp_modmul_modinv_vs_python():
    r = F.Rng("modexp")
    for bits in (2048, 4096):
        k32 = bits // 32
        mods = [r.bits(bits) | (1 << (bits - 1)) | 1 for _ in range(3)] + [r.bits(bits - 7) | 1]
        B = 8
        idx = [i % len(mods) for i in range(B)]
        base = [r.bits(bits) for _ in range(B)]
        exp = [r.bits(300) for _ in range(B)]
        exp[0] = 0
        base[1] = 0
        got = F.ints(orc.modexp(F.words(mods, k32), F.words(base, k32), F.words(exp, 10), idx))
        assert got == [pow(b, e, mods[i]) for b, e, i in zip(base, exp, idx)]
        b2 = [r.bits(bits) for _ in range(B)]
        got = F.ints(orc.modmul(F.words(mods, k32), F.words(base, k32), F.words(b2, k32), idx))
        assert got == [x * y % mods[i] for x, y, i in zip(base, b2, idx)]
        inv, ok = orc.modinv(F.words(mods, k32), F.words(b2, k32), idx)
        for v, o, x, i in zip(F.ints(inv), ok, b2, idx):
            try:
                want = pow(x, -1, mods[i])
            except ValueError:
                want = None
            assert (o == 1 and v == want) if want is not None else o == 0


def test_paillier_vs_python(keys):
    r = F.Rng("paillier")
    B = 6
    kidx = [i % 4 for i in range(B)]
    N = F.words([k.N for k in keys[:4]], 64)
    m = [r.below(pyref.Q) if i % 2 == 0 else r.below(keys[kidx[i]].N) for i in range(B)]
    rr = [r.below(keys[kidx[i]].N) for i in range(B)]
    c = orc.paillier_encrypt(N, F.words(m, 64), F.words(rr, 64), kidx)
    assert F.ints(c) == [pyref.paillier_encrypt(keys[k].N, mm, x) for k, mm, x in zip(kidx, m, rr)]
    p = F.words([k.p for k in keys[:4]], 32)
    q = F.words([k.q for k in keys[:4]], 32)
    dec = F.ints(orc.paillier_decrypt(p, q, c, kidx))
    assert dec == m
    assert dec == [pyref.paillier_decrypt_textbook(keys[k].p, keys[k].q, cc) for k, cc in zip(kidx, F.ints(c))]
    # homomorphic ops: Dec(c1*c2) = m1+m2, Dec(c^k) = k*m   (mta/mod.rs:140-145)
    c2 = orc.paillier_encrypt(N, F.words(m[::-1], 64), F.words(rr[::-1], 64), kidx)
    s = F.ints(orc.paillier_decrypt(p, q, orc.paillier_add(N, c, c2, kidx), kidx))
    assert s == [(a + b) % keys[k].N for a, b, k in zip(m, m[::-1], kidx)]
    kk = [r.below(pyref.Q) for _ in range(B)]
    s = F.ints(orc.paillier_decrypt(p, q, orc.paillier_mul(N, c, F.words(kk, 64), kidx), kidx))
    assert s == [a * b % keys[k].N for a, b, k in zip(m, kk, kidx)]


def _alice_case(keys, r, i):
    ek, st = keys[i % 4], keys[4 + i % 3]
    a = r.below(pyref.Q)
    rr = r.below(ek.N)
    c = pyref.paillier_encrypt(ek.N, a, rr)
    return ek, st, a, rr, c, F.alice_nonces(r, ek, st)


def test_alice_proof_vs_python_and_roundtrip(keys):
    """range_proofs.rs:615-634 (generate -> verify) + bit-exact agreement oracle vs pyref"""
    r = F.Rng("alice")
    B = 4
    cases = [_alice_case(keys, r, i) for i in range(B)]
    N = F.words([k.N for k in keys[:4]], 64)
    Nt, h1, h2 = (F.words([getattr(k, f) for k in keys[4:7]], 64) for f in ("Nt", "h1", "h2"))
    kidx, sidx = [i % 4 for i in range(B)], [i % 3 for i in range(B)]
    pr = orc.alice_generate(N, Nt, h1, h2, kidx, sidx, F.words([c[2] for c in cases], 8), F.words([c[4] for c in cases], 128),
                            F.words([c[3] for c in cases], 64), F.words([c[5]["alpha"] for c in cases], 24),
                            F.words([c[5]["beta"] for c in cases], 64), F.words([c[5]["gamma"] for c in cases], 88),
                            F.words([c[5]["rho"] for c in cases], 72))
    for i, (ek, st, a, rr, c, nn) in enumerate(cases):
        want = pyref.alice_generate(ek.N, st.Nt, st.h1, st.h2, a, c, rr, **nn)
        for f in ("z", "e", "s", "s1", "s2"):
            assert F.ints(pr[f])[i] == want[f], f
        assert pyref.alice_verify(ek.N, st.Nt, st.h1, st.h2, c, want)
    cw = F.words([c[4] for c in cases], 128)
    assert list(orc.alice_verify(N, Nt, h1, h2, kidx, sidx, cw, pr)) == [1] * B
    # negatives: tampered s, tampered ciphertext, s1 > q^3 (:118)
    bad = {k: v.copy() for k, v in pr.items()}
    bad["s"][0, 0] ^= 1
    bad["s1"][1] = F.words([pyref.Q ** 3 + 1], 25)[0]
    cw2 = cw.copy()
    cw2[2, 3] ^= 4
    assert list(orc.alice_verify(N, Nt, h1, h2, kidx, sidx, cw2, bad)) == [0, 0, 0, 1]


def test_pdl_with_slack_vs_python_and_soundness(keys):
    """zk_pdl_with_slack/test.rs:11-68 (prove -> verify) and :70-129 (x+1 encrypted -> reject)"""
    r = F.Rng("pdl")
    B = 3
    N = F.words([k.N for k in keys[:4]], 64)
    Nt, h1, h2 = (F.words([getattr(k, f) for k in keys[4:7]], 64) for f in ("Nt", "h1", "h2"))
    kidx, sidx = [i % 4 for i in range(B)], [i % 3 for i in range(B)]
    xs, rs, cs, Qs, Gs, nn = [], [], [], [], [], []
    for i in range(B):
        ek, st = keys[kidx[i]], keys[4 + sidx[i]]
        x, rr = r.below(pyref.Q), r.below(ek.N)
        Gp = pyref.ec_mul(r.below(pyref.Q), pyref.G)          # G = R, a variable base (party_i.rs:691-717)
        xs.append(x); rs.append(rr); Gs.append(Gp); Qs.append(pyref.ec_mul(x, Gp))
        cs.append(pyref.paillier_encrypt(ek.N, x + (1 if i == 2 else 0), rr))   # case 2: wrong plaintext
        nn.append(F.pdl_nonces(r, ek, st))
    pr = orc.pdl_prove(N, Nt, h1, h2, kidx, sidx, F.words(cs, 128), F.point_words(Qs), F.point_words(Gs), F.words(xs, 8),
                       F.words(rs, 64), F.words([n["alpha"] for n in nn], 24), F.words([n["beta"] for n in nn], 64),
                       F.words([n["rho"] for n in nn], 72), F.words([n["gamma"] for n in nn], 88))
    for i in range(B):
        ek, st = keys[kidx[i]], keys[4 + sidx[i]]
        want = pyref.pdl_prove(ek.N, st.Nt, st.h1, st.h2, cs[i], Qs[i], Gs[i], xs[i], rs[i], **nn[i])
        for f in ("z", "u2", "u3", "s1", "s2", "s3"):
            assert F.ints(pr[f])[i] == want[f], f
        assert F.points(pr["u1"])[i] == want["u1"]
        assert pyref.pdl_verify(ek.N, st.Nt, st.h1, st.h2, cs[i], Qs[i], Gs[i], want) == (i != 2)
    ok = orc.pdl_verify(N, Nt, h1, h2, kidx, sidx, F.words(cs, 128), F.point_words(Qs), F.point_words(Gs), pr)
    assert list(ok) == [1, 1, 0]


def test_bob_proof_vs_python_and_roundtrip(keys):
    """range_proofs.rs:636-709: generate(check=false)->verify(None); generate(check=true)->BobProofExt::verify"""
    r = F.Rng("bob")
    N = F.words([k.N for k in keys[:4]], 64)
    Nt, h1, h2 = (F.words([getattr(k, f) for k in keys[4:7]], 64) for f in ("Nt", "h1", "h2"))
    for check in (False, True):
        B = 2
        kidx, sidx = [i % 4 for i in range(B)], [i % 3 for i in range(B)]
        rows = []
        for i in range(B):
            ek, st = keys[kidx[i]], keys[4 + sidx[i]]
            a, b, bp, rr = r.below(pyref.Q), r.below(pyref.Q), r.below(ek.N), r.below(ek.N)
            a_enc = pyref.paillier_encrypt(ek.N, a, r.below(ek.N))
            mta = pow(a_enc, b, ek.NN) * pyref.paillier_encrypt(ek.N, bp, rr) % ek.NN
            rows.append((ek, st, a_enc, mta, b, bp, rr, F.bob_nonces(r, ek, st)))
        col = lambda j, w: F.words([x[j] for x in rows], w)
        nn = lambda f, w: F.words([x[7][f] for x in rows], w)
        pr, u = orc.bob_generate(N, Nt, h1, h2, kidx, sidx, col(2, 128), col(3, 128), col(4, 8), col(5, 64), col(6, 64),
                                 nn("alpha", 24), nn("beta", 64), nn("gamma", 80), nn("rho", 72), nn("rho_prim", 88),
                                 nn("sigma", 72), nn("tau", 88), check)
        Xs = []
        for i, (ek, st, a_enc, mta, b, bp, rr, nz) in enumerate(rows):
            want, wu = pyref.bob_generate(ek.N, st.Nt, st.h1, st.h2, a_enc, mta, b, bp, rr, check=check, **nz)
            for f in ("t", "z", "e", "s", "s1", "s2", "t1", "t2"):
                assert F.ints(pr[f])[i] == want[f], f
            X = pyref.ec_mul(b, pyref.G)
            Xs.append(X)
            if check:
                assert F.points(u)[i] == wu
                assert pyref.bob_verify(ek.N, st.Nt, st.h1, st.h2, a_enc, mta, want, X, wu)
            else:
                assert pyref.bob_verify(ek.N, st.Nt, st.h1, st.h2, a_enc, mta, want)
        Xw = F.point_words(Xs) if check else None
        assert list(orc.bob_verify(N, Nt, h1, h2, kidx, sidx, col(2, 128), col(3, 128), pr, Xw, u)) == [1, 1]
        bad = {k: v.copy() for k, v in pr.items()}
        bad["t2"][0, 1] ^= 8
        assert list(orc.bob_verify(N, Nt, h1, h2, kidx, sidx, col(2, 128), col(3, 128), bad, Xw, u)) == [0, 1]


def test_dlog_proof_vs_python():
    r = F.Rng("dlog")
    sk = [r.below(pyref.Q) for _ in range(3)]
    nonce = [r.below(pyref.Q) for _ in range(3)]
    pk, R, z = orc.dlog_prove(F.words(sk, 8), F.words(nonce, 8))
    for i in range(3):
        wpk, wR, wz = pyref.dlog_prove(sk[i], nonce[i])
        assert (F.points(pk)[i], F.points(R)[i], F.ints(z)[i]) == (wpk, wR, wz)
        assert pyref.dlog_verify(wpk, wR, wz)
    assert list(orc.dlog_verify(pk, R, z)) == [1, 1, 1]
    z[1, 0] ^= 1
    assert list(orc.dlog_verify(pk, R, z)) == [1, 0, 1]


def test_mta_identity(keys):
    """mta/test.rs:6-19: alpha + beta == a*b (mod q) through MessageA -> MessageB -> decrypt"""
    r = F.Rng("mta")
    ek = keys[0]
    a, b = r.below(pyref.Q), r.below(pyref.Q)
    N = F.words([ek.N], 64)
    c_a = orc.paillier_encrypt(N, F.words([a], 64), F.words([r.below(ek.N)], 64))
    beta_tag = r.below(ek.N)
    c_bt = orc.paillier_encrypt(N, F.words([beta_tag], 64), F.words([r.below(ek.N)], 64))
    c_b = orc.paillier_add(N, orc.paillier_mul(N, c_a, F.words([b], 64)), c_bt)
    alpha = F.ints(orc.paillier_decrypt(F.words([ek.p], 32), F.words([ek.q], 32), c_b))[0] % pyref.Q
    beta = (-beta_tag) % pyref.Q
    assert (alpha + beta) % pyref.Q == a * b % pyref.Q
