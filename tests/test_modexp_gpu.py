"""GPU parity tests (through the C-ABI): mpe_modexp / mpe_modmul vs the GMP oracle, the committed
golden vectors, and size-independent properties at BASELINE.json's full batch size."""
import json
import os

import numpy as np
import pytest
import torch

import fixtures as F
import orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
H = lambda s: int(s, 16)


def _engine():
    from multi_party_ecdsa_amd import engine
    return engine


def test_hip_library_is_loaded(gpu_ctx):
    from multi_party_ecdsa_amd import _native
    assert os.path.exists(_native.LIB_PATH)
    assert b"gfx950" in _native.lib.mpe_version()
    maps = open("/proc/self/maps").read()
    assert "libmpecdsa_hip.so" in maps


def test_modexp_golden_vectors(gpu_ctx):
    E = _engine()
    with open(os.path.join(HERE, "golden", "golden_small.json")) as f:
        gold = json.load(f)["modexp"]
    for bits in (2048, 4096):
        for ebits in sorted({g["ebits"] for g in gold if g["bits"] == bits}):
            gs = [g for g in gold if g["bits"] == bits and g["ebits"] == ebits]
            ms = E.ModSet(gpu_ctx, bits, [H(g["mod"]) for g in gs])
            got = E.mod_pow(gpu_ctx, ms, [H(g["base"]) for g in gs], [H(g["exp"]) for g in gs], exp_bits=ebits)
            assert got == [H(g["out"]) for g in gs]


@pytest.mark.parametrize("bits,ebits,B,nmod", [(2048, 256, 333, 7), (2048, 2816, 130, 16), (4096, 2048, 200, 16),
                                                (4096, 256, 257, 257), (4096, 769, 65, 3), (2048, 1024, 64, 2)])
def test_modexp_vs_oracle(gpu_ctx, keys, bits, ebits, B, nmod):
    """ragged batch sizes (not a multiple of the 8/16 integers a wave holds), shared and per-item moduli"""
    E = _engine()
    r = F.Rng(f"gpu-modexp-{bits}-{ebits}")
    k32 = bits // 32
    mods = [r.bits(bits) | (1 << (bits - 1)) | 1 for _ in range(nmod)]
    if nmod >= 3:
        mods[0] = keys[0].N ** (bits // 2048)          # a real Paillier modulus N (or N^2)
        mods[1] = r.bits(bits - 40) | 1                # short modulus
        mods[2] = (1 << bits) - 1
    idx = [(i * 7) % nmod for i in range(B)]
    base = [r.bits(bits) for _ in range(B)]
    exp = [r.bits(ebits) for _ in range(B)]
    base[0], base[1], base[2] = 0, 1, mods[idx[2]]      # edge bases
    exp[3], exp[4], exp[5] = 0, 1, (1 << ebits) - 1     # edge exponents
    ms = E.ModSet(gpu_ctx, bits, mods)
    got = E.mod_pow(gpu_ctx, ms, base, exp, mod_idx=idx, exp_bits=ebits)
    want = F.ints(orc.modexp(F.words(mods, k32), F.words(base, k32), F.words(exp, (ebits + 31) // 32), idx))
    bad = [i for i in range(B) if got[i] != want[i]]
    assert not bad, f"{len(bad)} mismatches, first {bad[:5]}"


@pytest.mark.parametrize("bits,ebits,e2bits,B,nmod", [(4096, 2048, 256, 150, 3), (2048, 2048, 256, 70, 5), (4096, 800, 256, 33, 33),
                                                       (2048, 320, 32, 40, 1)])
def test_modexp2_is_the_product_of_the_two_powers(gpu_ctx, keys, bits, ebits, e2bits, B, nmod):
    """mpe_modexp2 (one ladder, shared squarings) == mod_pow * mod_pow % n of the oracle, incl. zero windows and edge exponents"""
    E = _engine()
    r = F.Rng(f"gpu-modexp2-{bits}-{ebits}-{e2bits}")
    k32 = bits // 32
    mods = [r.bits(bits) | (1 << (bits - 1)) | 1 for _ in range(nmod)]
    if nmod >= 3:
        mods[0] = keys[0].N ** (bits // 2048)
    idx = [(i * 5) % nmod for i in range(B)]
    b1, b2 = [r.bits(bits) for _ in range(B)], [r.bits(bits) for _ in range(B)]
    e1, e2 = [r.bits(ebits) for _ in range(B)], [r.bits(e2bits) for _ in range(B)]
    e1[0], e2[0] = 0, 0
    e1[1], e2[1] = (1 << ebits) - 1, (1 << e2bits) - 1
    e2[2] = 0
    e1[3] = 0
    e2[4] = 0xF0F0F0F0 & ((1 << e2bits) - 1)
    b2[5] = 1
    ms = E.ModSet(gpu_ctx, bits, mods)
    got = E.mod_pow2(gpu_ctx, ms, b1, e1, b2, e2, mod_idx=idx, exp_bits=ebits, exp2_bits=e2bits)
    mw = F.words(mods, k32)
    p1 = orc.modexp(mw, F.words(b1, k32), F.words(e1, (ebits + 31) // 32), idx)
    p2 = orc.modexp(mw, F.words(b2, k32), F.words(e2, (e2bits + 31) // 32), idx)
    want = F.ints(orc.modmul(mw, p1, p2, idx))
    bad = [i for i in range(B) if got[i] != want[i]]
    assert not bad, f"{len(bad)} mismatches, first {bad[:5]}"


@pytest.mark.parametrize("bits", [2048, 4096])
def test_modmul_vs_oracle(gpu_ctx, bits):
    E = _engine()
    r = F.Rng(f"gpu-modmul-{bits}")
    k32, B, nmod = bits // 32, 301, 5
    mods = [r.bits(bits) | (1 << (bits - 1)) | 1 for _ in range(nmod)]
    mods[1] = 3
    idx = [i % nmod for i in range(B)]
    a = [r.bits(bits) for _ in range(B)]
    b = [r.bits(bits) for _ in range(B)]
    a[0], b[1] = 0, 0
    a[2], b[2] = (1 << bits) - 1, (1 << bits) - 1
    got = E.mod_mul(gpu_ctx, ms := E.ModSet(gpu_ctx, bits, mods), a, b, mod_idx=idx)
    want = F.ints(orc.modmul(F.words(mods, k32), F.words(a, k32), F.words(b, k32), idx))
    assert got == want
    assert ms.count == nmod


def test_empty_batch_and_bad_args(gpu_ctx):
    E = _engine()
    from multi_party_ecdsa_amd import _native as N
    ms = E.ModSet(gpu_ctx, 2048, [(1 << 2047) | 1])
    assert N.lib.mpe_modexp(gpu_ctx.h, ms.h, 0, None, 1, 1, 1, 1, None) == N.MPE_OK       # empty batch is a no-op
    assert N.lib.mpe_modexp(gpu_ctx.h, ms.h, 4, None, None, None, 1, None, None) == N.MPE_E_ARG
    h = N.C.c_void_p()
    assert N.lib.mpe_modset_create(gpu_ctx.h, 1024, 1, 1, N.C.byref(h), None) == N.MPE_E_ARG  # unsupported width


def test_full_size_properties(gpu_ctx, keys):
    """BASELINE.json config 2 size (65 536 x modulus 4096 / exponent 2048): size-independent checks.
    (1) x^(e1) * x^(e2) == x^(e1+e2) (mod N^2) for every item  [homomorphism of modexp]
    (2) a strided sample of 256 items is bit-exact against the oracle."""
    E = _engine()
    B, bits, k32 = 65536, 4096, 128
    ms = E.ModSet(gpu_ctx, bits, [k.NN for k in keys])
    g = torch.Generator(device="cuda")
    g.manual_seed(2024)
    dev = gpu_ctx.device
    base = torch.randint(-2**31, 2**31 - 1, (B, k32), dtype=torch.int32, device=dev, generator=g)
    e1 = torch.randint(0, 2**31 - 1, (B, 64), dtype=torch.int32, device=dev, generator=g)
    e2 = torch.randint(0, 2**31 - 1, (B, 64), dtype=torch.int32, device=dev, generator=g)
    e12 = e1 + e2                                            # every word < 2^31: no carries between words
    idx = (torch.arange(B, device=dev, dtype=torch.int32) % len(keys)).contiguous()
    y1 = E.modexp_device(gpu_ctx, ms, base, e1, d_mod_idx=idx)
    y2 = E.modexp_device(gpu_ctx, ms, base, e2, d_mod_idx=idx)
    y12 = E.modexp_device(gpu_ctx, ms, base, e12, d_mod_idx=idx)
    prod = E.modmul_device(gpu_ctx, ms, y1, y2, d_mod_idx=idx)
    gpu_ctx.sync()
    assert torch.equal(prod, y12)
    sel = torch.arange(0, B, 256, device=dev)
    hb, he, hy = (t[sel].cpu().numpy().view(np.uint32) for t in (base, e1, y1))
    want = orc.modexp(F.words([k.NN for k in keys], k32), np.ascontiguousarray(hb), np.ascontiguousarray(he),
                      [int(i) % len(keys) for i in sel.cpu()])
    assert np.array_equal(np.ascontiguousarray(hy), want)
