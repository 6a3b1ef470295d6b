"""ctypes binding of the CPU oracle (oracle/libmpe_oracle.so) for tests / bench cpu_baseline.
Arrays are numpy uint32 [batch, words] (same layout as the product's C-ABI)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libmpe_oracle.so")

W256, W768, WS1, W1024, W2048, W2304, W2560, WT1, W2816, WS2, W4096, WPOINT = 8, 24, 25, 32, 64, 72, 80, 81, 88, 89, 128, 16


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def _stale():
    """the library is missing or older than one of its sources"""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    src = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h")) and f != "ossl_check.c"]
    return any(os.path.getmtime(f) > t for f in src)


def load():
    # `make` is a no-op when the libraries are newer than their sources and rebuilds stale ones (a kept work tree).  A failed
    # build is only tolerated when the library on disk is NEWER than every source (e.g. a box without make): a stale oracle
    # must never be loaded silently.
    try:
        build()
    except (OSError, subprocess.CalledProcessError):
        if _stale():
            raise
    return C.CDLL(LIB)


lib = load()
lib.orc_version.restype = C.c_char_p


class Encoding(C.Structure):
    """orc_encoding (oracle/mpe_oracle.h) — field for field the product's mpe_encoding"""
    _fields_ = [("chain_point", C.c_uint8), ("zero_bytes", C.c_uint8), ("ck_mask_order", C.c_uint8), ("reserved", C.c_uint8),
                ("ck_salt", C.c_uint32), ("ord_dlog", C.c_uint8 * 4), ("ord_pedersen", C.c_uint8 * 8), ("ord_heg", C.c_uint8 * 8),
                ("ord_ecddh", C.c_uint8 * 8), ("ord_cdlog", C.c_uint8 * 4)]
    SIZES = dict(ord_dlog=3, ord_pedersen=5, ord_heg=7, ord_ecddh=6, ord_cdlog=4)


def get_encoding():
    e = Encoding()
    lib.orc_get_encoding(C.byref(e))
    out = {k: getattr(e, k) for k in ("chain_point", "zero_bytes", "ck_mask_order", "ck_salt")}
    out.update({k: list(getattr(e, k))[:n] for k, n in Encoding.SIZES.items()})
    return out


def set_encoding(d):
    """d: the dict form of a profile (pyref.Encoding.as_dict()); fields left out keep their current value.  Process-wide."""
    e = Encoding()
    lib.orc_get_encoding(C.byref(e))
    for k, v in d.items():
        if k.startswith("ord_"):
            arr = getattr(e, k)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            setattr(e, k, v)
    lib.orc_set_encoding(C.byref(e))


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _idx(idx):
    return None if idx is None else np.ascontiguousarray(idx, dtype=np.int32)


def u32(shape):
    return np.zeros(shape, dtype=np.uint32)


def modexp(mods, base, exp, mod_idx=None):
    B, k32 = base.shape
    out = u32((B, k32))
    mi = _idx(mod_idx)
    lib.orc_modexp(k32, B, mods.shape[0], _p(mods), _p(mi), _p(base), _p(exp), exp.shape[1], _p(out))
    return out


def modmul(mods, a, b, mod_idx=None):
    B, k32 = a.shape
    out = u32((B, k32))
    mi = _idx(mod_idx)
    lib.orc_modmul(k32, B, mods.shape[0], _p(mods), _p(mi), _p(a), _p(b), _p(out))
    return out


def modinv(mods, a, mod_idx=None):
    B, k32 = a.shape
    out, ok = u32((B, k32)), np.zeros(B, dtype=np.uint8)
    mi = _idx(mod_idx)
    lib.orc_modinv(k32, B, mods.shape[0], _p(mods), _p(mi), _p(a), _p(out), _p(ok))
    return out, ok


def paillier_encrypt(N, m, r, key_idx=None):
    B = m.shape[0]
    c = u32((B, W4096))
    ki = _idx(key_idx)
    lib.orc_paillier_encrypt(B, N.shape[0], _p(N), _p(ki), _p(m), _p(r), _p(c))
    return c


def paillier_decrypt(p, q, c, key_idx=None):
    B = c.shape[0]
    m = u32((B, W2048))
    ki = _idx(key_idx)
    lib.orc_paillier_decrypt(B, p.shape[0], _p(p), _p(q), _p(ki), _p(c), _p(m))
    return m


def paillier_add(N, c1, c2, key_idx=None):
    B = c1.shape[0]
    out = u32((B, W4096))
    ki = _idx(key_idx)
    lib.orc_paillier_add(B, N.shape[0], _p(N), _p(ki), _p(c1), _p(c2), _p(out))
    return out


def paillier_mul(N, c, k, key_idx=None):
    B = c.shape[0]
    out = u32((B, W4096))
    ki = _idx(key_idx)
    lib.orc_paillier_mul(B, N.shape[0], _p(N), _p(ki), _p(c), _p(k), _p(out))
    return out


def sha256(msg: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib.orc_sha256(msg, C.c_uint64(len(msg)), out)
    return bytes(out)


def ec_mul_base(k):
    out = u32((k.shape[0], WPOINT))
    lib.orc_ec_mul_base(k.shape[0], _p(k), _p(out))
    return out


def ec_mul(k, P):
    out = u32((k.shape[0], WPOINT))
    lib.orc_ec_mul(k.shape[0], _p(k), _p(P), _p(out))
    return out


def ec_add(P, Q):
    out = u32((P.shape[0], WPOINT))
    lib.orc_ec_add(P.shape[0], _p(P), _p(Q), _p(out))
    return out


def ec_compress(P):
    out = np.zeros((P.shape[0], 33), dtype=np.uint8)
    lib.orc_ec_compress(P.shape[0], _p(P), _p(out))
    return out


def alice_generate(N, Nt, h1, h2, key_idx, st_idx, a, cipher, r, alpha, beta, gamma, rho):
    B = a.shape[0]
    z, e, s, s1, s2 = u32((B, W2048)), u32((B, W256)), u32((B, W2048)), u32((B, WS1)), u32((B, WS2))
    ki, si = _idx(key_idx), _idx(st_idx)
    lib.orc_alice_generate(B, N.shape[0], _p(N), Nt.shape[0], _p(Nt), _p(h1), _p(h2), _p(ki), _p(si), _p(a), _p(cipher),
                           _p(r), _p(alpha), _p(beta), _p(gamma), _p(rho), _p(z), _p(e), _p(s), _p(s1), _p(s2))
    return dict(z=z, e=e, s=s, s1=s1, s2=s2)


def alice_verify(N, Nt, h1, h2, key_idx, st_idx, cipher, pr):
    B = cipher.shape[0]
    ok = np.zeros(B, dtype=np.uint8)
    ki, si = _idx(key_idx), _idx(st_idx)
    lib.orc_alice_verify(B, N.shape[0], _p(N), Nt.shape[0], _p(Nt), _p(h1), _p(h2), _p(ki), _p(si), _p(cipher),
                         _p(pr["z"]), _p(pr["e"]), _p(pr["s"]), _p(pr["s1"]), _p(pr["s2"]), _p(ok))
    return ok


def pdl_prove(N, Nt, h1, h2, key_idx, st_idx, cipher, Q, G, x, r, alpha, beta, rho, gamma):
    B = x.shape[0]
    o = dict(z=u32((B, W2048)), u1=u32((B, WPOINT)), u2=u32((B, W4096)), u3=u32((B, W2048)), s1=u32((B, WS1)),
             s2=u32((B, W2048)), s3=u32((B, WS2)))
    ki, si = _idx(key_idx), _idx(st_idx)
    lib.orc_pdl_prove(B, N.shape[0], _p(N), Nt.shape[0], _p(Nt), _p(h1), _p(h2), _p(ki), _p(si), _p(cipher), _p(Q), _p(G),
                      _p(x), _p(r), _p(alpha), _p(beta), _p(rho), _p(gamma), _p(o["z"]), _p(o["u1"]), _p(o["u2"]),
                      _p(o["u3"]), _p(o["s1"]), _p(o["s2"]), _p(o["s3"]))
    return o


def pdl_verify(N, Nt, h1, h2, key_idx, st_idx, cipher, Q, G, pr):
    B = cipher.shape[0]
    ok = np.zeros(B, dtype=np.uint8)
    ki, si = _idx(key_idx), _idx(st_idx)
    lib.orc_pdl_verify(B, N.shape[0], _p(N), Nt.shape[0], _p(Nt), _p(h1), _p(h2), _p(ki), _p(si), _p(cipher), _p(Q), _p(G),
                       _p(pr["z"]), _p(pr["u1"]), _p(pr["u2"]), _p(pr["u3"]), _p(pr["s1"]), _p(pr["s2"]), _p(pr["s3"]), _p(ok))
    return ok


def bob_generate(N, Nt, h1, h2, key_idx, st_idx, a_enc, mta_enc, b, beta_prim, r, alpha, beta, gamma, rho, rho_prim,
                 sigma, tau, check):
    B = b.shape[0]
    o = dict(t=u32((B, W2048)), z=u32((B, W2048)), e=u32((B, W256)), s=u32((B, W2048)), s1=u32((B, WS1)),
             s2=u32((B, WS2)), t1=u32((B, WT1)), t2=u32((B, WS2)))
    u = u32((B, WPOINT)) if check else None
    ki, si = _idx(key_idx), _idx(st_idx)
    lib.orc_bob_generate(B, N.shape[0], _p(N), Nt.shape[0], _p(Nt), _p(h1), _p(h2), _p(ki), _p(si), _p(a_enc), _p(mta_enc),
                         _p(b), _p(beta_prim), _p(r), _p(alpha), _p(beta), _p(gamma), _p(rho), _p(rho_prim), _p(sigma),
                         _p(tau), int(bool(check)), _p(o["t"]), _p(o["z"]), _p(o["e"]), _p(o["s"]), _p(o["s1"]),
                         _p(o["s2"]), _p(o["t1"]), _p(o["t2"]), _p(u))
    return o, u


def bob_verify(N, Nt, h1, h2, key_idx, st_idx, a_enc, mta_enc, pr, X=None, u=None):
    B = a_enc.shape[0]
    ok = np.zeros(B, dtype=np.uint8)
    ki, si = _idx(key_idx), _idx(st_idx)
    lib.orc_bob_verify(B, N.shape[0], _p(N), Nt.shape[0], _p(Nt), _p(h1), _p(h2), _p(ki), _p(si), _p(a_enc), _p(mta_enc),
                       _p(pr["t"]), _p(pr["z"]), _p(pr["e"]), _p(pr["s"]), _p(pr["s1"]), _p(pr["s2"]), _p(pr["t1"]),
                       _p(pr["t2"]), _p(X), _p(u), _p(ok))
    return ok


def dlog_prove(sk, nonce):
    B = sk.shape[0]
    pk, R, z = u32((B, WPOINT)), u32((B, WPOINT)), u32((B, W256))
    lib.orc_dlog_prove(B, _p(sk), _p(nonce), _p(pk), _p(R), _p(z))
    return pk, R, z


def dlog_verify(pk, R, z):
    ok = np.zeros(pk.shape[0], dtype=np.uint8)
    lib.orc_dlog_verify(pk.shape[0], _p(pk), _p(R), _p(z), _p(ok))
    return ok


def lindell_partial_sig(N, c_key, x2, k2, R1, msg, rho, r, key_idx=None):
    """PartialSig::compute (lindell_2017/party_two.rs:390-423): c3 [B][128]"""
    B = c_key.shape[0]
    c3 = u32((B, W4096))
    lib.orc_lindell_partial_sig(B, N.shape[0], _p(N), _p(_idx(key_idx)), _p(c_key), _p(x2), _p(k2), _p(R1), _p(msg), _p(rho),
                                _p(r), _p(c3))
    return c3


def lindell_sign(p, q, c3, k1, R2, key_idx=None):
    """Signature::compute_with_recid (lindell_2017/party_one.rs:519-565): r, s [B][8], recid [B]"""
    B = c3.shape[0]
    r, s, recid = u32((B, 8)), u32((B, 8)), np.zeros(B, dtype=np.int32)
    lib.orc_lindell_sign(B, p.shape[0], _p(p), _p(q), _p(_idx(key_idx)), _p(c3), _p(k1), _p(R2), _p(r), _p(s), _p(recid))
    return r, s, recid


# ---- the sampler's restatement (oracle/sampler_oracle.c): curv Samplable / from_modulo / Scalar::random over ChaCha20 ----
lib.orc_sample_bits.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
lib.orc_sample_below.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.orc_sample_scalar.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_void_p]
lib.orc_sampler_set_max_attempts.argtypes = [C.c_int]
lib.orc_sampler_set_max_attempts.restype = None
lib.orc_chacha20_block.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p]
lib.orc_gg20_sample_nonces.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
SAMPLE_NONZERO, SAMPLE_PLUS_ONE, SAMPLE_COPRIME = 1, 2, 4


def chacha20_block(key, counter, n13, n14, n15):
    out = C.create_string_buffer(64)
    lib.orc_chacha20_block(bytes(key), counter, n13, n14, n15, out)
    return out.raw


lib.orc_sample_rule.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.orc_sample_rule.restype = None


def sample_rule(buf, bits, out_words):
    """curv's BigInt::sample(bits) on GIVEN bytes (from_bytes_be >> (8 len - bits)), as the sampler oracle applies it"""
    out = u32((1, out_words))
    lib.orc_sample_rule(bytes(buf), len(buf), bits, out_words, _p(out))
    return out


def sample_bits(batch, seed, sid, bits, out_words):
    out = u32((batch, out_words))
    lib.orc_sample_bits(batch, bytes(seed), sid, bits, out_words, _p(out))
    return out


def sample_below(batch, seed, sid, bound, out_words, bound_idx=None, flags=0):
    """bound: uint32 [nbounds, bound_words]; returns (values, failures)"""
    out = u32((batch, out_words))
    fails = lib.orc_sample_below(batch, bytes(seed), sid, _p(bound), bound.shape[1], bound.shape[0], _p(_idx(bound_idx)), flags, out_words, _p(out))
    return out, fails


def sample_scalar(batch, seed, sid):
    out = u32((batch, 8))
    fails = lib.orc_sample_scalar(batch, bytes(seed), sid, _p(out))
    return out, fails
