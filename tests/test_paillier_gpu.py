"""GPU parity tests: batched Paillier encrypt / decrypt / add / mul through the C-ABI vs the GMP
oracle and the golden vectors; BASELINE.json config 2 (65 536 ops, 16 keys) via round-trip and
homomorphism properties."""
import json
import os

import numpy as np
import pytest
import torch

import fixtures as F
import orc
import pyref

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
H = lambda s: int(s, 16)


@pytest.fixture(scope="module")
def pk(gpu_ctx, keys):
    from multi_party_ecdsa_amd import engine as E
    return E.PaillierKeys(gpu_ctx, p=[k.p for k in keys], q=[k.q for k in keys])


def test_private_keyset_derives_n(pk, keys, gpu_ctx):
    from multi_party_ecdsa_amd import _native as N
    import ctypes
    ptr = N.lib.mpe_paillier_n(pk.h)
    t = torch.empty((len(keys), 64), dtype=torch.int32, device=gpu_ctx.device)
    # read the device table through torch: copy from raw pointer
    buf = (ctypes.c_uint32 * (len(keys) * 64)).from_address  # noqa: F841 (documentation of the layout)
    c = pk.encrypt([0] * len(keys), [1] * len(keys), list(range(len(keys))))      # Enc(0; r=1) = 1
    assert c == [1] * len(keys)
    c = pk.encrypt([1] * len(keys), [1] * len(keys), list(range(len(keys))))      # Enc(1; r=1) = 1 + N
    assert c == [1 + k.N for k in keys]
    assert ptr and t.shape[0] == pk.nkeys


def test_paillier_golden(pk, keys):
    with open(os.path.join(HERE, "golden", "golden_small.json")) as f:
        gold = json.load(f)["paillier"]
    kidx = [g["key"] for g in gold]
    c = pk.encrypt([H(g["m"]) for g in gold], [H(g["r"]) for g in gold], kidx)
    assert c == [H(g["c"]) for g in gold]
    assert pk.decrypt(c, kidx) == [H(g["m"]) for g in gold]


def test_encrypt_decrypt_vs_oracle_ragged(pk, keys):
    r = F.Rng("gpu-paillier")
    B = 203                                       # not a multiple of 8 or 16
    kidx = [(3 * i) % len(keys) for i in range(B)]
    m = [r.below(pyref.Q) if i % 2 == 0 else r.below(keys[kidx[i]].N) for i in range(B)]   # k_i-like / beta'-like
    rr = [r.below(keys[kidx[i]].N) for i in range(B)]
    m[0], m[1], rr[2] = 0, keys[kidx[1]].N - 1, 1                                         # edges
    c = pk.encrypt(m, rr, kidx)
    N = F.words([k.N for k in keys], 64)
    want = F.ints(orc.paillier_encrypt(N, F.words(m, 64), F.words(rr, 64), kidx))
    assert c == want
    got_m = pk.decrypt(c, kidx)
    want_m = F.ints(orc.paillier_decrypt(F.words([k.p for k in keys], 32), F.words([k.q for k in keys], 32),
                                         F.words(c, 128), kidx))
    assert got_m == want_m == m


def test_add_mul_vs_oracle_and_mta_identity(pk, keys):
    """Paillier::mul / add as used by MessageB::b (mta/mod.rs:140-145), then alpha + beta = a*b (mta/test.rs:16-18)"""
    r = F.Rng("gpu-mta")
    B = 37
    kidx = [i % len(keys) for i in range(B)]
    a = [r.below(pyref.Q) for _ in range(B)]
    b = [r.below(pyref.Q) for _ in range(B)]
    bt = [r.below(keys[k].N) for k in kidx]
    c_a = pk.encrypt(a, [r.below(keys[k].N) for k in kidx], kidx)
    c_bt = pk.encrypt(bt, [r.below(keys[k].N) for k in kidx], kidx)
    b_c_a = pk.mul(c_a, b, kidx)
    c_b = pk.add(b_c_a, c_bt, kidx)
    N = F.words([k.N for k in keys], 64)
    assert b_c_a == F.ints(orc.paillier_mul(N, F.words(c_a, 128), F.words(b, 64), kidx))
    assert c_b == F.ints(orc.paillier_add(N, F.words(b_c_a, 128), F.words(c_bt, 128), kidx))
    alpha = [x % pyref.Q for x in pk.decrypt(c_b, kidx)]
    assert [(al + (-t) % pyref.Q) % pyref.Q for al, t in zip(alpha, bt)] == [x * y % pyref.Q for x, y in zip(a, b)]


def test_public_keyset_and_errors(gpu_ctx, keys):
    from multi_party_ecdsa_amd import engine as E, _native as N
    pub = E.PaillierKeys(gpu_ctx, N=[k.N for k in keys[:3]])
    r = F.Rng("gpu-pub")
    m, rr = [r.below(pyref.Q) for _ in range(3)], [r.below(k.N) for k in keys[:3]]
    assert pub.encrypt(m, rr) == [pyref.paillier_encrypt(k.N, x, y) for k, x, y in zip(keys, m, rr)]   # key_idx None -> key i
    with pytest.raises(N.MpeError):
        pub.decrypt([1, 2, 3])                    # no private part: rejected, not silently wrong


def test_config2_full_size_roundtrip(pk, keys, gpu_ctx):
    """BASELINE config 2: 65 536 encrypt + decrypt, 16 keys.  Decrypt(Encrypt(m, r)) == m for every item
    (round trip), a strided 128-item sample of the ciphertexts is bit-exact against the oracle."""
    B = 65536
    dev = gpu_ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    m = torch.zeros((B, 64), dtype=torch.int32, device=dev)
    m[:, :8] = torch.randint(-2**31, 2**31 - 1, (B, 8), dtype=torch.int32, device=dev, generator=g)      # 256-bit (k_i case)
    wide = torch.randint(-2**31, 2**31 - 1, (B // 2, 63), dtype=torch.int32, device=dev, generator=g)   # < 2^2016 < N
    m[B // 2:, :63] = wide
    rr = torch.zeros((B, 64), dtype=torch.int32, device=dev)
    rr[:, :63] = torch.randint(-2**31, 2**31 - 1, (B, 63), dtype=torch.int32, device=dev, generator=g)
    idx = (torch.arange(B, device=dev, dtype=torch.int32) % len(keys)).contiguous()
    c = pk.encrypt_device(m, rr, idx)
    back = pk.decrypt_device(c, idx)
    gpu_ctx.sync()
    assert torch.equal(back, m)
    sel = torch.arange(0, B, 512, device=dev)
    want = orc.paillier_encrypt(F.words([k.N for k in keys], 64), np.ascontiguousarray(m[sel].cpu().numpy().view(np.uint32)),
                                np.ascontiguousarray(rr[sel].cpu().numpy().view(np.uint32)), [int(i) % len(keys) for i in sel.cpu()])
    assert np.array_equal(np.ascontiguousarray(c[sel].cpu().numpy().view(np.uint32)), want)


@pytest.mark.parametrize("env", [{"no_pair": 1}, {"no_crt": 1}, {"no_pown": 1}, {"no_pair": 1, "no_crt": 1},
                                 {"window_bits": 4}, {"window_bits": 5}, {"no_adaptive_lanes": 1}])
def test_every_arithmetic_route_gives_the_same_ciphertexts(pk, keys, env):
    """The A/B switches select different algorithms for the same residues (N-adic pairs vs the 4096-bit kernel, the
    holder's p^2|q^2 halves, x^N through a^p, window widths): all of them must agree with the default route, which
    the other tests pin to the oracle."""
    from multi_party_ecdsa_amd import engine as E
    r = F.Rng("gpu-routes")
    B = 37
    kidx = [(5 * i) % len(keys) for i in range(B)]
    m = [r.below(keys[kidx[i]].N) for i in range(B)]
    rr = [r.below(keys[kidx[i]].N) for i in range(B)]
    want = pk.encrypt(m, rr, kidx)
    ctx2 = E.Context(0, options=env)                         # mpe_ctx_set_option: the library itself reads no environment
    sk2 = E.PaillierKeys(ctx2, p=[k.p for k in keys], q=[k.q for k in keys])
    pk2 = E.PaillierKeys(ctx2, N=[k.N for k in keys])
    assert sk2.encrypt(m, rr, kidx) == want                  # key holder
    assert pk2.encrypt(m, rr, kidx) == want                  # public key only
    assert sk2.decrypt(want, kidx) == m


@pytest.mark.parametrize("kw", [8, 25, 64])
def test_pair_engine_edge_operands(pk, keys, kw):
    """c^k mod N^2 through the N-adic pair kernel at the corners: bases 0, 1, N, N^2 - 1, 2^4096 - 1 (unreduced), a
    multiple of p; exponents 0, 1, all-ones; every exponent width class (4-, 5- and 6-bit windows)."""
    r = F.Rng(f"gpu-pair-edges-{kw}")
    nk = len(keys)
    bases, exps, kidx = [], [], []
    for i in range(24):
        k = keys[i % nk]
        kidx.append(i % nk)
        bases.append([0, 1, k.N, k.N * k.N - 1, (1 << 4096) - 1, k.p * 12345, r.below(k.N * k.N), r.bits(4096)][i % 8])
        exps.append([0, 1, (1 << (32 * kw)) - 1, r.bits(32 * kw)][(i // 8 + i) % 4])
    got = pk.mul(bases, exps, kidx, k_words=kw)
    assert got == [pow(b, e, keys[j].N ** 2) for b, e, j in zip(bases, exps, kidx)]


def test_decrypt_edge_ciphertexts(pk, keys):
    """ciphertexts 1, 1 + N, N^2 - 1 and an unreduced one (c + N^2 < 2^4096) decrypt like the oracle says"""
    nk = len(keys)
    cs, kidx = [], []
    for i in range(4 * nk):
        k = keys[i % nk]
        kidx.append(i % nk)
        cs.append([1, 1 + k.N, k.N * k.N - 1, 1 + 5 * k.N + (k.N * k.N if (k.N * k.N).bit_length() < 4096 else 0)][i // nk])
    got = pk.decrypt(cs, kidx)
    want = F.ints(orc.paillier_decrypt(F.words([k.p for k in keys], 32), F.words([k.q for k in keys], 32), F.words(cs, 128), kidx))
    assert got == want
    assert got[:nk] == [0] * nk and got[nk:2 * nk] == [1] * nk and got[2 * nk:3 * nk] == [0] * nk and got[3 * nk:] == [5] * nk


def test_holder_paths_on_degenerate_randomness(pk, keys, gpu_ctx):
    """r = 1, N - 1, multiples of p and of q (not units: the a^p shortcut must not rely on Fermat there), r just below
    N: the key holder's p^2 | q^2 route and the public route give the oracle's ciphertext."""
    from multi_party_ecdsa_amd import engine as E
    pub = E.PaillierKeys(gpu_ctx, N=[k.N for k in keys])
    nk = len(keys)
    m, rr, kidx = [], [], []
    for i in range(5 * nk):
        k = keys[i % nk]
        kidx.append(i % nk)
        m.append((i * 0x9E3779B97F4A7C15) % k.N)
        rr.append([1, k.N - 1, k.p * 3, k.q * (k.p - 1), k.p * k.p % k.N][i // nk])
    want = F.ints(orc.paillier_encrypt(F.words([k.N for k in keys], 64), F.words(m, 64), F.words(rr, 64), kidx))
    assert pk.encrypt(m, rr, kidx) == want
    assert pub.encrypt(m, rr, kidx) == want


@pytest.mark.parametrize("B", [1, 15, 16, 17, 333])
def test_public_exponent_sliding_windows_equal_fixed_windows_and_the_oracle(gpu_ctx, keys, B):
    """x^N for the PUBLIC exponent N runs on sliding windows with the launch ordered by key (mpe_pairexp.h); a context created
    with option no_sliding keeps the fixed windows.  Peer-side encryption (r^N mod N^2) and the two-base MessageB ciphertext
    c_a^b r^N over 16 keys in an order that makes every wave straddle key boundaries: both contexts and the GMP oracle agree
    bit for bit, at batch sizes around the wave's 16 groups."""
    from multi_party_ecdsa_amd import engine as E
    ctx_fixed = E.Context(0, options={"no_sliding": 1})
    r = F.Rng(f"sliding-{B}")
    kidx = [(5 * i + i // 7) % len(keys) for i in range(B)]
    m = [r.below(keys[k].N) for k in kidx]
    rr = [r.below(keys[k].N) for k in kidx]
    rr[0] = 1                                                    # edge bases
    if B > 1:
        rr[1] = keys[kidx[1]].N - 1
    N = F.words([k.N for k in keys], 64)
    want = F.ints(orc.paillier_encrypt(N, F.words(m, 64), F.words(rr, 64), kidx))
    outs = []
    b = [r.below(pyref.Q) for _ in range(B)]
    for ctx in (gpu_ctx, ctx_fixed):
        pub = E.PaillierKeys(ctx, N=[k.N for k in keys])
        assert pub.encrypt(m, rr, kidx) == want
        # MessageB::b's ciphertext: c_a^b * Enc(beta'; r) — the short SECRET exponent b stays on fixed windows
        c_a = want
        got = pub.add(pub.mul(c_a, b, kidx), pub.encrypt(m, rr, kidx), kidx)
        ref = [pow(ca, bb, keys[k].N ** 2) * w % keys[k].N ** 2 for ca, bb, w, k in zip(c_a, b, want, kidx)]
        assert got == ref
        outs.append(got)
    assert outs[0] == outs[1]
    ctx_fixed.close()


def test_sliding_windows_on_sparse_and_dense_public_exponents(gpu_ctx):
    """the window scan on exponents an RSA modulus never looks like: long runs of zeros (a window every few hundred bits), all
    ones (back-to-back full windows), a lone top bit above a low word — r^N mod N^2 against Python's pow, 20 items per 'key'
    so that every wave holds one key and the sliding schedule really runs"""
    from multi_party_ecdsa_amd import engine as E
    mods = [(1 << 2047) + (1 << 1000) + 1, (1 << 2048) - 159, (1 << 2047) + 0xF00F, (1 << 2047) + (1 << 1536) + (1 << 1024) + (1 << 512) + 3,
            ((1 << 2048) - 1) // 3 | (1 << 2047) | 1]
    pub = E.PaillierKeys(gpu_ctx, N=mods)
    r = F.Rng("sliding-sparse")
    kidx = [i % len(mods) for i in range(20 * len(mods))]
    rr = [r.below(mods[k]) for k in kidx]
    m = [r.below(mods[k]) for k in kidx]
    got = pub.encrypt(m, rr, kidx)
    want = [(1 + mm * mods[k]) * pow(x, mods[k], mods[k] ** 2) % mods[k] ** 2 for mm, x, k in zip(m, rr, kidx)]
    assert got == want
