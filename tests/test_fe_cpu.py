"""The device field / point code (csrc/mpe_fe.h, csrc/mpe_jac.h) compiled for the HOST and fuzzed against Python
integers and tests/pyref.py: the 10 x 26-bit lazy-reduction limb algorithms are checked at every magnitude the point
formulas use, at the worst-case limb patterns, and through whole scalar multiplications (variable base, comb table,
exceptional additions).  No GPU involved: the same header text is what hipcc compiles for gfx950."""
import ctypes
import os
import random
import subprocess

import pytest

import pyref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = R.P
M26 = (1 << 26) - 1


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(ROOT, "build", "libfe_host.so")
    src = os.path.join(ROOT, "tools", "model", "fe_host.cpp")
    inc = os.path.join(ROOT, "multi_party_ecdsa_amd", "csrc")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    deps = [src, os.path.join(inc, "mpe_fe.h"), os.path.join(inc, "mpe_jac.h"), os.path.join(inc, "mpe_sc.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-DMPE_FE_HOST", "-I", inc, src, "-o", out])
    return ctypes.CDLL(out)


def arr(vals, n):
    return (ctypes.c_uint32 * n)(*vals)


def limbs_value(l):
    return sum(int(v) << (26 * i) for i, v in enumerate(l))


def words_value(w):
    return sum(int(v) << (32 * i) for i, v in enumerate(w))


def to_words(x, n=8):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def rand_limbs(rng, m, mode):
    """limbs of an element of magnitude m: n[0..8] <= 2 m (2^26 - 1), n[9] <= 2 m (2^22 - 1)"""
    hi = [2 * m * M26] * 9 + [2 * m * ((1 << 22) - 1)]
    if mode == "max":
        return hi
    if mode == "edge":
        return [rng.choice([0, 1, h - 1, h, h // 2]) for h in hi]
    return [rng.randrange(h + 1) for h in hi]


def test_mul_sqr_all_magnitudes(lib):
    rng = random.Random(1)
    out, ol = arr([0] * 8, 8), arr([0] * 10, 10)
    pairs = [(1, 1), (1, 3), (3, 3), (3, 6), (2, 6), (5, 5), (5, 3), (1, 32), (4, 8), (2, 16), (3, 2)]
    for m1, m2 in pairs:
        for mode in ["max", "edge", "rand", "rand", "rand", "edge", "rand"] * 6:
            a, b = rand_limbs(rng, m1, mode), rand_limbs(rng, m2, mode if mode != "edge" else "rand")
            lib.feh_mul(arr(a, 10), arr(b, 10), out, ol)
            assert words_value(out) == limbs_value(a) * limbs_value(b) % P, (m1, m2, mode)
            # the raw result is of magnitude 1 and congruent
            assert all(v <= 2 * M26 for v in ol[:9]) and ol[9] <= 2 * ((1 << 22) - 1)
            assert limbs_value(ol) % P == words_value(out)
    for m in [1, 2, 3, 5]:
        for mode in ["max", "edge", "rand", "rand", "edge", "rand"] * 8:
            a = rand_limbs(rng, m, mode)
            lib.feh_sqr(arr(a, 10), out, ol)
            assert words_value(out) == limbs_value(a) ** 2 % P, (m, mode)
            assert all(v <= 2 * M26 for v in ol[:9]) and ol[9] <= 2 * ((1 << 22) - 1)


def test_normalize_weak_zero_neg(lib):
    rng = random.Random(2)
    out, ol = arr([0] * 8, 8), arr([0] * 10, 10)
    specials = [0, 1, P - 1, P, P + 1, 2 * P, 2 * P - 1, (1 << 256) - 1, 1 << 256, (1 << 256) + (1 << 32) + 976, 3 * P, 31 * P]
    for v in specials:
        # the value spread over limbs in a few ways (canonical split, and with a heavy top limb)
        l0 = [(v >> (26 * i)) & M26 for i in range(9)] + [v >> 234]
        for l in [l0]:
            if max(l) >= 1 << 32:
                continue
            lib.feh_normalize(arr(l, 10), out)
            assert words_value(out) == v % P, hex(v)
            assert lib.feh_is_zero(arr(l, 10)) == (1 if v % P == 0 else 0), hex(v)
    for m in [1, 2, 3, 6, 10, 17, 31]:
        for mode in ["max", "edge", "rand", "rand", "rand"] * 10:
            a = rand_limbs(rng, m, mode)
            lib.feh_normalize(arr(a, 10), out)
            assert words_value(out) == limbs_value(a) % P
            lib.feh_weak(arr(a, 10), ol)
            assert limbs_value(ol) % P == limbs_value(a) % P
            assert all(v <= M26 for v in ol[:9]) and ol[9] <= (1 << 22) + 63
            assert lib.feh_is_zero(arr(a, 10)) == (1 if limbs_value(a) % P == 0 else 0)
            if m < 31:
                lib.feh_neg(arr(a, 10), m, ol)
                assert (limbs_value(ol) + limbs_value(a)) % P == 0
                hi = [2 * (m + 1) * M26] * 9 + [2 * (m + 1) * ((1 << 22) - 1)]
                assert all(0 <= v <= h for v, h in zip(ol, hi))
    # multiples of p at every magnitude are recognised as zero
    for k in range(0, 60):
        v = k * P
        l = [(v >> (26 * i)) & M26 for i in range(9)] + [v >> 234]
        assert lib.feh_is_zero(arr(l, 10)) == 1


def test_words_roundtrip_and_inverse(lib):
    rng = random.Random(3)
    out, ol = arr([0] * 8, 8), arr([0] * 10, 10)
    for v in [0, 1, P - 1, (1 << 255) + 12345, 0xDEADBEEF << 200] + [rng.randrange(P) for _ in range(50)]:
        lib.feh_from_words(arr(to_words(v), 8), ol)
        assert limbs_value(ol) == v and all(x <= M26 for x in ol)
        lib.feh_normalize(ol, out)
        assert words_value(out) == v
        if v:
            for k in [1, 2, 5]:
                lk = [x * k for x in ol]
                lib.feh_inv(arr(lk, 10), out)
                assert words_value(out) * v * k % P == 1


def pt_words(pt):
    return [0] * 16 if pt is None else to_words(pt[0]) + to_words(pt[1])


def words_pt(w):
    x, y = words_value(w[:8]), words_value(w[8:])
    return None if x == 0 and y == 0 else (x, y)


def test_scalar_multiplication_and_additions(lib):
    rng = random.Random(4)
    out = arr([0] * 16, 16)
    pts = [R.G, R.H2] + [R.ec_mul(rng.randrange(1, R.Q), R.G) for _ in range(6)]
    ks = [0, 1, 2, 3, 15, 16, 17, R.Q - 1, R.Q - 2, (1 << 255), (1 << 256) - 1 - (1 << 256) % 1, 0x1111111111111111111111111111111111111111111111111111111111111111,
          0xF0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0 % R.Q] + [rng.randrange(R.Q) for _ in range(12)]
    for pt in pts[:4]:
        for k in ks:
            k %= R.Q
            lib.ech_mul(arr(to_words(k), 8), arr(pt_words(pt), 16), out)
            assert words_pt(out) == R.ec_mul(k, pt), hex(k)
    # mixed additions incl. the exceptional cases: P + P, P + (-P), P + inf, inf + P
    for a in pts[:4]:
        for b in [a, R.ec_neg(a), None, pts[5], R.ec_mul(2, a)]:
            lib.ech_add(arr(pt_words(a), 16), arr(pt_words(b), 16), out)
            assert words_pt(out) == R.ec_add(a, b)
            lib.ech_add(arr(pt_words(None), 16), arr(pt_words(b), 16), out)
            assert words_pt(out) == b
    # general Jacobian addition of two computed points, incl. equal, opposite and infinite operands
    cases = [(3, pts[2], 5, pts[3]), (7, R.G, 7, R.G), (7, R.G, R.Q - 7, R.G), (0, R.G, 9, pts[4]), (9, pts[4], 0, R.G),
             (0, R.G, 0, R.G), (2, R.G, 1, R.ec_mul(2, R.G))] + [(rng.randrange(R.Q), pts[2], rng.randrange(R.Q), pts[6]) for _ in range(6)]
    for ka, pa, kb, pb in cases:
        lib.ech_add_jac(arr(to_words(ka), 8), arr(pt_words(pa), 16), arr(to_words(kb), 8), arr(pt_words(pb), 16), out)
        assert words_pt(out) == R.ec_add(R.ec_mul(ka, pa), R.ec_mul(kb, pb))
    # projective equality
    for k in [1, 2, 5, rng.randrange(R.Q)]:
        q = R.ec_mul(k, pts[3])
        assert lib.ech_eq(arr(to_words(k), 8), arr(pt_words(pts[3]), 16), arr(pt_words(q), 16)) == 3
        assert lib.ech_eq(arr(to_words(k), 8), arr(pt_words(pts[3]), 16), arr(pt_words(R.ec_neg(q)), 16)) == 0
        assert lib.ech_eq(arr(to_words(k), 8), arr(pt_words(pts[3]), 16), arr(pt_words(pts[4]), 16)) == 0
    assert lib.ech_eq(arr(to_words(0), 8), arr(pt_words(R.G), 16), arr(pt_words(None), 16)) == 3
    # curve membership
    for pt in pts:
        assert lib.ech_on_curve(arr(pt_words(pt), 16)) == 1
        assert lib.ech_on_curve(arr(pt_words((pt[0], (pt[1] + 1) % P)), 16)) == 0
        assert lib.ech_on_curve(arr(pt_words(((pt[0] + 1) % P, pt[1])), 16)) == 0


def test_comb_tables(lib):
    rng = random.Random(5)
    out = arr([0] * 16, 16)
    for base in [R.G, R.H2]:
        tab = (ctypes.c_uint32 * (64 * 15 * 20))()
        lib.ech_comb_build(arr(pt_words(base), 16), tab)
        # spot-check entries, then multiply
        for w, d in [(0, 1), (0, 15), (1, 1), (63, 15), (31, 7)]:
            e = tab[(w * 15 + d - 1) * 20:(w * 15 + d) * 20]
            assert (limbs_value(e[:10]), limbs_value(e[10:])) == R.ec_mul(d * 16 ** w, base)
        for k in [0, 1, 15, 16, R.Q - 1, 1 << 255 % R.Q, 0x0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F] + [rng.randrange(R.Q) for _ in range(10)]:
            k %= R.Q
            lib.ech_mul_comb(arr(to_words(k), 8), tab, out)
            assert words_pt(out) == R.ec_mul(k, base), hex(k)


def test_scalar_field(lib):
    rng = random.Random(6)
    Q = R.Q
    out = arr([0] * 8, 8)
    edge = [0, 1, 2, Q - 1, Q, Q + 1, (1 << 256) - 1, (1 << 255), (1 << 128) - 1, 1 << 128, Q >> 1]
    # reductions of every width the kernels use (8 .. 90 words), incl. all-ones inputs and multiples of q
    for n in [1, 5, 8, 9, 16, 24, 25, 64, 72, 88, 89, 90]:
        vals = [0, (1 << (32 * n)) - 1, (Q * ((1 << (32 * n)) // Q)) if n >= 8 else 0, ((1 << (32 * n)) // Q) * Q - 1 if n >= 8 else 1]
        vals += [rng.randrange(1 << (32 * n)) for _ in range(12)]
        for v in vals:
            lib.sch_reduce(arr(to_words(v, n), n), n, out)
            assert words_value(out) == v % Q, (n, hex(v))
    for a in edge:
        for b in edge + [rng.randrange(1 << 256) for _ in range(3)]:
            lib.sch_mul(arr(to_words(a % (1 << 256)), 8), arr(to_words(b % (1 << 256)), 8), out)
            assert words_value(out) == (a % (1 << 256)) * (b % (1 << 256)) % Q
    for _ in range(300):
        a, b = rng.randrange(1 << 256), rng.randrange(1 << 256)
        lib.sch_mul(arr(to_words(a), 8), arr(to_words(b), 8), out)
        assert words_value(out) == a * b % Q
    for a in [1, 2, Q - 1, Q - 2] + [rng.randrange(1, Q) for _ in range(20)]:
        lib.sch_inv(arr(to_words(a), 8), out)
        assert words_value(out) * a % Q == 1


LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
BETA = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE


def test_glv_constants_and_split(lib):
    """the endomorphism constants are re-derived here from the lattice basis; the device split is exact and 128-bit"""
    n, p = R.Q, R.P
    assert pow(LAMBDA, 3, n) == 1 and LAMBDA != 1 and pow(BETA, 3, p) == 1 and BETA != 1
    assert R.ec_mul(LAMBDA, R.G) == (BETA * R.G[0] % p, R.G[1])
    a1, b1 = 0x3086D221A7D46BCDE86C90E49284EB15, -0xE4437ED6010E88286F547FA90ABFE4C3
    a2, b2 = 0x114CA50F7A8E2F3F657C1108D9D44CFD8, a1
    assert (a1 + b1 * LAMBDA) % n == 0 and (a2 + b2 * LAMBDA) % n == 0
    g1, g2 = (2 * (b2 << 384) + n) // (2 * n), (2 * ((-b1) << 384) + n) // (2 * n)
    src = open(os.path.join(ROOT, "multi_party_ecdsa_amd", "csrc", "mpe_sc.h")).read()

    def const(name):
        body = src[src.index(name + "[8] = {") + len(name) + 7:]
        return words_value([int(t.strip().rstrip("u"), 16) for t in body[:body.index("}")].split(",")])
    assert const("GLV_LAMBDA") == LAMBDA and const("GLV_BETA") == BETA
    assert const("GLV_G1") == g1 and const("GLV_G2") == g2
    assert const("GLV_MB1") == (-b1) % n and const("GLV_MB2") == (-b2) % n
    rng = random.Random(7)
    r1, r2, neg = arr([0] * 8, 8), arr([0] * 8, 8), (ctypes.c_int * 2)()
    ks = [0, 1, 2, n - 1, n - 2, n // 2, n // 2 + 1, LAMBDA, n - LAMBDA, LAMBDA + 1, (1 << 128), (1 << 128) - 1, (1 << 255)] + [rng.randrange(n) for _ in range(3000)]
    for k in ks:
        lib.sch_split(arr(to_words(k), 8), r1, r2, neg)
        v1, v2 = words_value(r1), words_value(r2)
        assert v1 < 1 << 128 and v2 < 1 << 128
        assert ((-v1 if neg[0] else v1) + (-v2 if neg[1] else v2) * LAMBDA) % n == k


def test_glv_ladder_matches_plain_ladder(lib):
    rng = random.Random(8)
    out, out2 = arr([0] * 16, 16), arr([0] * 16, 16)
    n = R.Q
    pts = [R.G, R.H2, R.ec_mul(rng.randrange(1, n), R.G)]
    ks = [0, 1, 2, 15, 16, 17, 31, 32, 33, n - 1, n - 2, LAMBDA, n - LAMBDA, LAMBDA - 1, LAMBDA + 1, (LAMBDA * 16) % n, (LAMBDA * 17 + 16) % n,
          (1 << 128) - 1, 1 << 128, 0x10842108421084210842108421084210 % n] + [rng.randrange(n) for _ in range(60)]
    for pt in pts:
        for k in ks:
            lib.ech_mul(arr(to_words(k), 8), arr(pt_words(pt), 16), out)
            lib.ech_mul_w4(arr(to_words(k), 8), arr(pt_words(pt), 16), out2)
            assert list(out) == list(out2), hex(k)
            assert words_pt(out) == R.ec_mul(k, pt), hex(k)
