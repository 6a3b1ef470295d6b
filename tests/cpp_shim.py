"""Builds tests/cpp/test_shim.cpp — the reference's own unit tests re-stated over include/mpecdsa.hpp, the C++ host layer above
the C-ABI — and writes the fixture file it reads (the seeded fixtures of tests/fixtures.py as flat little-endian word arrays).
Test infrastructure (the binary links the oracle and libgmp; the product library links neither)."""
import os
import struct
import subprocess

import numpy as np

import fixtures as F
import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(out_dir):
    exe = os.path.join(out_dir, "test_shim")
    lib, orc = os.path.join(ROOT, "multi_party_ecdsa_amd", "libmpecdsa_hip.so"), os.path.join(ROOT, "oracle", "libmpe_oracle.so")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    gmp = next(p for p in ("/opt/conda/lib/libgmp.so", "/usr/lib/x86_64-linux-gnu/libgmp.so.10") if os.path.exists(p))
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
           "-I", "/opt/conda/include", "-I", os.path.join(rocm, "include"), os.path.join(ROOT, "tests", "cpp", "test_shim.cpp"), "-o", exe,
           lib, orc, gmp, os.path.join(rocm, "lib", "libamdhip64.so"), "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + os.path.dirname(orc),
           "-Wl,-rpath," + os.path.join(rocm, "lib")]
    subprocess.check_call(cmd)
    return exe


def write_fixture(path, keys, B=6, seed="cpp-shim"):
    r = F.Rng(seed)
    rs = lambda: r.below(pyref.Q - 1) + 1
    ek, st = keys[:2], keys[5:7]
    nst = len(st)
    kidx = [i % 2 for i in range(B)]
    sidx = [(i + 1) % nst for i in range(B)]
    a, b = [rs() for _ in range(B)], [rs() for _ in range(B)]
    r_a = [r.coprime_below(ek[k].N) for k in kidx]
    one = [F.alice_nonces(r, ek[kidx[i]], st[sidx[i]]) for i in range(B)]                         # alice_zkp: one statement per item
    al = [F.alice_nonces(r, ek[kidx[i]], st[s]) for i in range(B) for s in range(nst)]            # MessageA: every statement per exchange
    pd = [F.pdl_nonces(r, ek[kidx[i]], st[sidx[i]]) for i in range(B)]
    arrays = {
        "N": F.words([k.N for k in ek], 64), "p": F.words([k.p for k in ek], 32), "q": F.words([k.q for k in ek], 32),
        "Nt": F.words([k.Nt for k in st], 64), "h1": F.words([k.h1 for k in st], 64), "h2": F.words([k.h2 for k in st], 64),
        "key_idx": np.array(kidx, dtype=np.uint32).reshape(B, 1), "st_idx": np.array(sidx, dtype=np.uint32).reshape(B, 1),
        "a": F.words(a, 8), "b": F.words(b, 8), "r_a": F.words(r_a, 64),
        "mb_r": F.words([r.coprime_below(ek[k].N) for k in kidx], 64), "beta_tag": F.words([r.below(ek[k].N) for k in kidx], 64),
        "nonce_b": F.words([rs() for _ in range(B)], 8), "nonce_bt": F.words([rs() for _ in range(B)], 8),
    }
    for f, w in (("alpha", 24), ("beta", 64), ("gamma", 88), ("rho", 72)):
        arrays["one_" + f] = F.words([n[f] for n in one], w)
        arrays["al_" + f] = F.words([n[f] for n in al], w)
        arrays["pdl_" + f] = F.words([n[f] for n in pd], w)
    with open(path, "wb") as f:
        for name, arr in arrays.items():
            arr = np.ascontiguousarray(arr, dtype=np.uint32)
            f.write(struct.pack("<I", len(name)) + name.encode() + struct.pack("<II", arr.shape[1], arr.shape[0]) + arr.tobytes())
    return arrays
