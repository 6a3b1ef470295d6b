"""ctypes binding of the OpenSSL checker (oracle/libmpe_ossl.so: batch glue over libcrypto 1.1.1l's EC_POINT_mul,
ECDSA_do_verify, SHA256, BN_mod_exp) — third-party code as the independent pin of the EC / ECDSA / hash layer, the role
libsecp256k1 has in the reference's own `check_sig` (gg_2020/test.rs:711-748).  Test infrastructure: used by tests/ and by
bench.py AFTER the timed region.  Arrays are numpy uint32 in the C-ABI's word layout."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import orc

LIB = os.path.join(orc.ROOT, "oracle", "libmpe_ossl.so")
lib = C.CDLL(LIB)                      # orc's import has run `make` (both libraries are targets of oracle/Makefile)
lib.ossl_version.restype = C.c_char_p


def version():
    return lib.ossl_version().decode()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _chunks(n, threads):
    return [c for c in np.array_split(np.arange(n), max(1, min(threads, n))) if len(c)]


def ec_mul(k, P=None, threads=1):
    """k [B,8] scalars, P [B,16] points (None: the generator) -> [B,16]; OpenSSL EC_POINT_mul"""
    k = _c(k)
    P = None if P is None else _c(P)
    out = np.zeros((k.shape[0], 16), dtype=np.uint32)

    def run(ix):
        a, b = int(ix[0]), int(ix[-1]) + 1
        bad = lib.ossl_ec_mul(b - a, _p(k[a:b]), _p(None if P is None else P[a:b]), _p(out[a:b]))
        assert bad == 0, "OpenSSL rejected a point"
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, _chunks(k.shape[0], threads)))
    return out


def ec_add(P, Q):
    P, Q = _c(P), _c(Q)
    out = np.zeros_like(P)
    assert lib.ossl_ec_add(P.shape[0], _p(P), _p(Q), _p(out)) == 0
    return out


def ecdsa_verify(pub, msg, r, s, threads=1):
    """pub [16] (one key) or [B,16]; msg, r, s [B,8] -> bool [B]; OpenSSL ECDSA_do_verify with digest = the message"""
    pub, msg, r, s = _c(pub), _c(msg), _c(r), _c(s)
    B = msg.shape[0]
    stride = 0 if pub.ndim == 1 else 16
    ok = np.zeros(B, dtype=np.uint8)

    def run(ix):
        a, b = int(ix[0]), int(ix[-1]) + 1
        lib.ossl_ecdsa_verify(b - a, _p(pub if stride == 0 else pub[a:b]), stride, _p(msg[a:b]), _p(r[a:b]), _p(s[a:b]), _p(ok[a:b]))
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, _chunks(B, threads)))
    return ok.astype(bool)


def sha256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib.ossl_sha256(data, C.c_size_t(len(data)), out)
    return out.raw


def modexp(base: int, exp: int, mod: int) -> int:
    ml = (mod.bit_length() + 7) // 8
    b, e, m = base.to_bytes(max(1, (base.bit_length() + 7) // 8), "big"), exp.to_bytes(max(1, (exp.bit_length() + 7) // 8), "big"), mod.to_bytes(ml, "big")
    out = C.create_string_buffer(ml)
    assert lib.ossl_modexp(b, len(b), e, len(e), m, ml, out) == 0
    return int.from_bytes(out.raw, "big")
