"""Thin object layer over the C-ABI: device buffers are torch tensors (int32 storage of the
u32 words), every compute call goes to libmpecdsa_hip.so.  Mirrors curv's `BigInt::mod_pow` /
`mod_mul` in batched form."""
import ctypes as C
import os

import numpy as np
import torch

from . import _native as N
N_ = N
from .words import ints_to_words, words_to_ints


def _dev_u32(arr_np, device):
    """np.uint32 [.., ..] -> torch int32 tensor on device holding the same bits."""
    return torch.from_numpy(arr_np.view(np.int32)).to(device)


def _to_np_u32(t):
    return t.detach().cpu().numpy().view(np.uint32)


# The LIBRARY reads no environment variable (mpe_ctx_set_option is the only way to change an A/B switch).  This harness — tests,
# bench.py, tools/ — keeps the MPE_* variables of the earlier rounds' measurement scripts working by translating them into options
# when it creates a context: MPE_NO_PAR=1 -> ("no_par", "1"), MPE_WIDE_DIV=4 -> ("wide_div", "4"), MPE_GRID=full -> ("grid", "full").
_ENV_OPTIONS = ["no_fixed_base", "no_crt", "no_multiexp", "no_pair", "no_pown", "no_sliding", "no_par", "no_wide", "no_adaptive_lanes",
                "no_merge_xn", "no_merge_r1", "fb_window_bits", "window_bits", "wide_div", "xwide_div", "waves_per_cu", "grid", "fb_budget_mb",
                "fb_split", "gg20_trace", "sampler_max_attempts", "no_elect", "no_primaries", "merge_r1_quarters",
                "no_r1_inversion_ahead", "no_r1_dlog_first", "no_prio", "no_pdl_ahead", "no_crt_n"]


def options_from_env(env=None):
    env = os.environ if env is None else env
    out = {}
    for k in _ENV_OPTIONS:
        v = env.get("MPE_" + k.upper())
        if v is not None and v != "":
            out[k] = v
    if env.get("MPE_GRID_EQUAL"):
        out["grid"] = "equal"
    return out


class Context:
    def __init__(self, device=0, encoding=None, options=None):
        """encoding: None (the defaults of include/mpecdsa_hip.h) or a dict / N.Encoding — the profile of the recalled
        curv / zk-paillier byte conventions this context hashes with (mpe_ctx_set_encoding).
        options: {key: value} for mpe_ctx_set_option, applied on top of the harness's MPE_* environment translation."""
        if not torch.cuda.is_available():
            raise N.MpeError("no GPU visible: the HIP path cannot run (there is no CPU fallback)")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        N.check(N.lib.mpe_ctx_create(C.byref(h), device), "mpe_ctx_create")
        self.h = h
        if encoding is not None:
            self.set_encoding(encoding)
        self.options = {}
        for k, v in {**options_from_env(), **(options or {})}.items():
            self.set_option(k, v)

    def set_option(self, key, value):
        """mpe_ctx_set_option: an A/B switch of the measurements (none changes a result); before the dependent objects are created"""
        N.check(N.lib.mpe_ctx_set_option(self.h, str(key).encode(), str(int(value) if isinstance(value, bool) else value).encode()), f"mpe_ctx_set_option({key})")
        self.options[key] = value

    def get_option(self, key):
        v = C.c_long(0)
        N.check(N.lib.mpe_ctx_get_option(self.h, str(key).encode(), C.byref(v)), f"mpe_ctx_get_option({key})")
        return v.value

    def set_device_share(self, contexts):
        """this many contexts work on the device at the same time (mpe_ctx_set_device_share): keep the efficient lane layouts"""
        N.check(N.lib.mpe_ctx_set_device_share(self.h, int(contexts)), "mpe_ctx_set_device_share")

    def set_encoding(self, encoding):
        e = encoding if isinstance(encoding, N.Encoding) else N.Encoding.from_dict(dict(encoding))
        N.check(N.lib.mpe_ctx_set_encoding(self.h, C.byref(e)), "mpe_ctx_set_encoding")

    def encoding(self):
        e = N.Encoding()
        N.check(N.lib.mpe_ctx_get_encoding(self.h, C.byref(e)), "mpe_ctx_get_encoding")
        return e.as_dict()

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def sync(self):
        N.check(N.lib.mpe_sync(self.h, self.stream()), "mpe_sync")

    def launch_info(self):
        li = N.LaunchInfo()
        N.check(N.lib.mpe_last_launch_info(self.h, C.byref(li)), "mpe_last_launch_info")
        return {k: getattr(li, k) for k, _ in li._fields_}

    def prof_enable(self, on=True):
        N.check(N.lib.mpe_prof_enable(self.h, int(on)), "mpe_prof_enable")

    def prof_collect(self, max_records=4096):
        arr = (N.ProfRec * max_records)()
        n = C.c_int(0)
        N.check(N.lib.mpe_prof_collect(self.h, arr, max_records, C.byref(n)), "mpe_prof_collect")
        return [dict(kind=r.kind, bits=r.bits, exp_words=r.exp_words, batch=r.batch, ms=r.ms, exp2_words=r.exp2_words,
                     sliding_frac=r.sliding_frac) for r in arr[:n.value]]

    def wipe(self):
        N.check(N.lib.mpe_ctx_wipe(self.h, self.stream()), "mpe_ctx_wipe")

    def scratch_audit(self):
        """(non-zero 32-bit words, total bytes) over every scratch region the context owns"""
        nz, tot = C.c_uint64(0), C.c_uint64(0)
        N.check(N.lib.mpe_ctx_scratch_audit(self.h, C.byref(nz), C.byref(tot), self.stream()), "mpe_ctx_scratch_audit")
        return nz.value, tot.value

    def close(self):
        if self.h:
            N.lib.mpe_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ModSet:
    """A set of odd moduli resident in HBM with their Montgomery constants (computed on the GPU)."""

    def __init__(self, ctx, bits, moduli):
        """moduli: list of ints, or a device int32 tensor [count, bits/32] of u32 words."""
        self.ctx, self.bits, self.k32 = ctx, bits, bits // 32
        if isinstance(moduli, torch.Tensor):
            self.d_moduli = moduli.contiguous()
        else:
            self.d_moduli = _dev_u32(ints_to_words(moduli, self.k32), ctx.device)
        self.count = self.d_moduli.shape[0]
        h = C.c_void_p()
        N.check(N.lib.mpe_modset_create(ctx.h, bits, self.count, C.c_void_p(self.d_moduli.data_ptr()),
                                        C.byref(h), ctx.stream()), "mpe_modset_create")
        self.h = h

    def close(self):
        if self.h:
            N.lib.mpe_modset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def modexp_device(ctx, ms, d_base, d_exp, d_out=None, d_mod_idx=None):
    """All-device call: d_base [B,k32], d_exp [B,ew], optional d_mod_idx [B] int32 -> d_out [B,k32]."""
    B = d_base.shape[0]
    if d_out is None:
        d_out = torch.empty_like(d_base)
    idx_ptr = C.c_void_p(d_mod_idx.data_ptr()) if d_mod_idx is not None else None
    N.check(N.lib.mpe_modexp(ctx.h, ms.h, B, idx_ptr, C.c_void_p(d_base.data_ptr()),
                             C.c_void_p(d_exp.data_ptr()), d_exp.shape[1], C.c_void_p(d_out.data_ptr()),
                             ctx.stream()), "mpe_modexp")
    return d_out


def modmul_device(ctx, ms, d_a, d_b, d_out=None, d_mod_idx=None):
    B = d_a.shape[0]
    if d_out is None:
        d_out = torch.empty_like(d_a)
    idx_ptr = C.c_void_p(d_mod_idx.data_ptr()) if d_mod_idx is not None else None
    N.check(N.lib.mpe_modmul(ctx.h, ms.h, B, idx_ptr, C.c_void_p(d_a.data_ptr()), C.c_void_p(d_b.data_ptr()),
                             C.c_void_p(d_out.data_ptr()), ctx.stream()), "mpe_modmul")
    return d_out


def mod_pow2(ctx, ms, bases, exps, bases2, exps2, mod_idx=None, exp_bits=None, exp2_bits=None):
    """Batched `mod_pow(b, e, n) * mod_pow(b2, e2, n) % n` on one ladder (`mpe_modexp2`); e2 is the short exponent."""
    ew = ((exp_bits or max(1, max(int(e).bit_length() for e in exps))) + 31) // 32
    ew2 = ((exp2_bits or max(1, max(int(e).bit_length() for e in exps2))) + 31) // 32
    dv = lambda xs, w: _dev_u32(ints_to_words(xs, w), ctx.device)
    d_idx = torch.tensor(mod_idx, dtype=torch.int32, device=ctx.device) if mod_idx is not None else None
    d_b, d_e, d_b2, d_e2 = dv(bases, ms.k32), dv(exps, ew), dv(bases2, ms.k32), dv(exps2, ew2)
    d_out = torch.empty_like(d_b)
    N.check(N.lib.mpe_modexp2(ctx.h, ms.h, len(bases), C.c_void_p(d_idx.data_ptr()) if d_idx is not None else None,
                              C.c_void_p(d_b.data_ptr()), C.c_void_p(d_e.data_ptr()), ew, C.c_void_p(d_b2.data_ptr()),
                              C.c_void_p(d_e2.data_ptr()), ew2, C.c_void_p(d_out.data_ptr()), ctx.stream()), "mpe_modexp2")
    ctx.sync()
    return words_to_ints(_to_np_u32(d_out))


def mod_pow(ctx, ms, bases, exps, mod_idx=None, exp_bits=None):
    """Batched `BigInt::mod_pow(base, exp, modulus)` on Python ints (host convenience wrapper)."""
    if exp_bits is None:
        exp_bits = max(1, max(int(e).bit_length() for e in exps))
    ew = (exp_bits + 31) // 32
    d_base = _dev_u32(ints_to_words(bases, ms.k32), ctx.device)
    d_exp = _dev_u32(ints_to_words(exps, ew), ctx.device)
    d_idx = torch.tensor(mod_idx, dtype=torch.int32, device=ctx.device) if mod_idx is not None else None
    d_out = modexp_device(ctx, ms, d_base, d_exp, d_mod_idx=d_idx)
    ctx.sync()
    return words_to_ints(_to_np_u32(d_out))


def mod_mul(ctx, ms, a, b, mod_idx=None):
    d_a = _dev_u32(ints_to_words(a, ms.k32), ctx.device)
    d_b = _dev_u32(ints_to_words(b, ms.k32), ctx.device)
    d_idx = torch.tensor(mod_idx, dtype=torch.int32, device=ctx.device) if mod_idx is not None else None
    d_out = modmul_device(ctx, ms, d_a, d_b, d_mod_idx=d_idx)
    ctx.sync()
    return words_to_ints(_to_np_u32(d_out))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class PaillierKeys:
    """A set of Paillier-2048 keys in HBM (`EncryptionKey{n,nn}` / `DecryptionKey{p,q}` of kzen-paillier).
    Build from moduli (public) or from primes (private; enables decrypt)."""

    def __init__(self, ctx, N=None, p=None, q=None):
        self.ctx = ctx
        h = C.c_void_p()
        if p is not None:
            self.d_p = _dev_u32(ints_to_words(p, 32), ctx.device)
            self.d_q = _dev_u32(ints_to_words(q, 32), ctx.device)
            N_.check(N_.lib.mpe_paillier_create_private(ctx.h, len(p), _ptr(self.d_p), _ptr(self.d_q), C.byref(h),
                                                        ctx.stream()), "mpe_paillier_create_private")
            self.private = True
        else:
            self.d_N = _dev_u32(ints_to_words(N, 64), ctx.device)
            N_.check(N_.lib.mpe_paillier_create_public(ctx.h, len(N), _ptr(self.d_N), C.byref(h), ctx.stream()),
                     "mpe_paillier_create_public")
            self.private = False
        self.h = h
        self.nkeys = N_.lib.mpe_paillier_nkeys(h)

    def close(self):
        if self.h:
            N_.lib.mpe_paillier_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- device-tensor API (int32 tensors holding u32 words) ----
    def encrypt_device(self, d_m, d_r, d_key_idx=None, d_c=None):
        """`Paillier::encrypt_with_chosen_randomness` batched: m,r [B,64] -> c [B,128]"""
        B = d_m.shape[0]
        if d_c is None:
            d_c = torch.empty((B, 128), dtype=torch.int32, device=d_m.device)
        N_.check(N_.lib.mpe_paillier_encrypt(self.ctx.h, self.h, B, _ptr(d_key_idx), _ptr(d_m), _ptr(d_r), _ptr(d_c),
                                             self.ctx.stream()), "mpe_paillier_encrypt")
        return d_c

    def decrypt_device(self, d_c, d_key_idx=None, d_m=None):
        """`Paillier::decrypt` batched: c [B,128] -> m [B,64]"""
        B = d_c.shape[0]
        if d_m is None:
            d_m = torch.empty((B, 64), dtype=torch.int32, device=d_c.device)
        N_.check(N_.lib.mpe_paillier_decrypt(self.ctx.h, self.h, B, _ptr(d_key_idx), _ptr(d_c), _ptr(d_m),
                                             self.ctx.stream()), "mpe_paillier_decrypt")
        return d_m

    def add_device(self, d_c1, d_c2, d_key_idx=None):
        out = torch.empty_like(d_c1)
        N_.check(N_.lib.mpe_paillier_add(self.ctx.h, self.h, d_c1.shape[0], _ptr(d_key_idx), _ptr(d_c1), _ptr(d_c2),
                                         _ptr(out), self.ctx.stream()), "mpe_paillier_add")
        return out

    def mul_device(self, d_c, d_k, d_key_idx=None):
        out = torch.empty_like(d_c)
        N_.check(N_.lib.mpe_paillier_mul(self.ctx.h, self.h, d_c.shape[0], _ptr(d_key_idx), _ptr(d_c), _ptr(d_k),
                                         d_k.shape[1], _ptr(out), self.ctx.stream()), "mpe_paillier_mul")
        return out

    # ---- Python-int convenience wrappers ----
    def _idx(self, key_idx):
        return None if key_idx is None else torch.tensor(key_idx, dtype=torch.int32, device=self.ctx.device)

    def encrypt(self, m, r, key_idx=None):
        d = self.encrypt_device(_dev_u32(ints_to_words(m, 64), self.ctx.device),
                                _dev_u32(ints_to_words(r, 64), self.ctx.device), self._idx(key_idx))
        self.ctx.sync()
        return words_to_ints(_to_np_u32(d))

    def decrypt(self, c, key_idx=None):
        d = self.decrypt_device(_dev_u32(ints_to_words(c, 128), self.ctx.device), self._idx(key_idx))
        self.ctx.sync()
        return words_to_ints(_to_np_u32(d))

    def add(self, c1, c2, key_idx=None):
        d = self.add_device(_dev_u32(ints_to_words(c1, 128), self.ctx.device),
                            _dev_u32(ints_to_words(c2, 128), self.ctx.device), self._idx(key_idx))
        self.ctx.sync()
        return words_to_ints(_to_np_u32(d))

    def mul(self, c, k, key_idx=None, k_words=8):
        d = self.mul_device(_dev_u32(ints_to_words(c, 128), self.ctx.device),
                            _dev_u32(ints_to_words(k, k_words), self.ctx.device), self._idx(key_idx))
        self.ctx.sync()
        return words_to_ints(_to_np_u32(d))


# ================================================================================================
# secp256k1, modinv, DLogStatement tables and the proofs (device-tensor API; int32 tensors of u32 words)
# ================================================================================================
def _new(ctx, B, words):
    return torch.empty((B, words), dtype=torch.int32, device=ctx.device)


def _flags(ctx, B):
    return torch.empty((B,), dtype=torch.uint8, device=ctx.device)


def dev(ctx, vals, words):
    """Python ints -> device word tensor"""
    return _dev_u32(ints_to_words(vals, words), ctx.device)


def host(t):
    """device word tensor -> Python ints"""
    return words_to_ints(_to_np_u32(t))


def modinv_device(ctx, ms, d_a, d_mod_idx=None):
    B = d_a.shape[0]
    out, ok = torch.empty_like(d_a), _flags(ctx, B)
    N_.check(N_.lib.mpe_modinv(ctx.h, ms.h, B, _ptr(d_mod_idx), _ptr(d_a), _ptr(out), _ptr(ok), ctx.stream()), "mpe_modinv")
    return out, ok


def ec_mul_base(ctx, d_k):
    out = _new(ctx, d_k.shape[0], 16)
    N_.check(N_.lib.mpe_ec_mul_base(ctx.h, d_k.shape[0], _ptr(d_k), d_k.shape[1], _ptr(out), ctx.stream()), "mpe_ec_mul_base")
    return out


def ec_mul(ctx, d_k, d_P):
    out = _new(ctx, d_k.shape[0], 16)
    N_.check(N_.lib.mpe_ec_mul(ctx.h, d_k.shape[0], _ptr(d_k), d_k.shape[1], _ptr(d_P), _ptr(out), ctx.stream()), "mpe_ec_mul")
    return out


def ec_add(ctx, d_P, d_Q):
    out = _new(ctx, d_P.shape[0], 16)
    N_.check(N_.lib.mpe_ec_add(ctx.h, d_P.shape[0], _ptr(d_P), _ptr(d_Q), _ptr(out), ctx.stream()), "mpe_ec_add")
    return out


def dlog_prove(ctx, d_sk, d_nonce):
    B = d_sk.shape[0]
    pk, R, z = _new(ctx, B, 16), _new(ctx, B, 16), _new(ctx, B, 8)
    N_.check(N_.lib.mpe_dlog_prove(ctx.h, B, _ptr(d_sk), _ptr(d_nonce), _ptr(pk), _ptr(R), _ptr(z), ctx.stream()), "mpe_dlog_prove")
    return pk, R, z


def dlog_verify(ctx, d_pk, d_R, d_z):
    ok = _flags(ctx, d_pk.shape[0])
    N_.check(N_.lib.mpe_dlog_verify(ctx.h, d_pk.shape[0], _ptr(d_pk), _ptr(d_R), _ptr(d_z), _ptr(ok), ctx.stream()), "mpe_dlog_verify")
    return ok


class Statements:
    """Table of `DLogStatement{N: N~, g: h1, ni: h2}` (party_i.rs:225-229) resident in HBM."""

    def __init__(self, ctx, Nt, h1, h2):
        self.ctx = ctx
        self.d = [dev(ctx, v, 64) for v in (Nt, h1, h2)]
        h = C.c_void_p()
        N_.check(N_.lib.mpe_statements_create(ctx.h, len(Nt), _ptr(self.d[0]), _ptr(self.d[1]), _ptr(self.d[2]),
                                              C.byref(h), ctx.stream()), "mpe_statements_create")
        self.h, self.count = h, len(Nt)

    def close(self):
        if self.h:
            N_.lib.mpe_statements_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


ALICE_PROOF_WORDS = dict(z=64, e=8, s=64, s1=25, s2=89)
ALICE_NONCE_WORDS = dict(alpha=24, beta=64, gamma=88, rho=72)
PDL_PROOF_WORDS = dict(z=64, u1=16, u2=128, u3=64, s1=25, s2=64, s3=89)
PDL_NONCE_WORDS = dict(alpha=24, beta=64, rho=72, gamma=88)


def _struct(cls, tensors):
    s = cls()
    for f, _ in cls._fields_:
        setattr(s, f, tensors[f].data_ptr())
    return s


def alice_generate(ctx, pk, stm, d_a, d_cipher, d_r, nonces, d_key_idx=None, d_st_idx=None):
    """`AliceProof::generate` batched.  nonces / result: dict of device tensors (widths: ALICE_*_WORDS)."""
    B = d_a.shape[0]
    out = {f: _new(ctx, B, w) for f, w in ALICE_PROOF_WORDS.items()}
    nn, pr = _struct(N_.AliceNonces, nonces), _struct(N_.AliceProof, out)
    N_.check(N_.lib.mpe_alice_generate(ctx.h, pk.h, stm.h, B, _ptr(d_key_idx), _ptr(d_st_idx), _ptr(d_a), _ptr(d_cipher),
                                       _ptr(d_r), C.byref(nn), C.byref(pr), ctx.stream()), "mpe_alice_generate")
    return out


def alice_verify(ctx, pk, stm, d_cipher, proof, d_key_idx=None, d_st_idx=None):
    B = d_cipher.shape[0]
    ok = _flags(ctx, B)
    pr = _struct(N_.AliceProof, proof)
    N_.check(N_.lib.mpe_alice_verify(ctx.h, pk.h, stm.h, B, _ptr(d_key_idx), _ptr(d_st_idx), _ptr(d_cipher), C.byref(pr),
                                     _ptr(ok), ctx.stream()), "mpe_alice_verify")
    return ok


def pdl_prove(ctx, pk, stm, d_cipher, d_Q, d_G, d_x, d_r, nonces, d_key_idx=None, d_st_idx=None):
    """`PDLwSlackProof::prove` batched."""
    B = d_x.shape[0]
    out = {f: _new(ctx, B, w) for f, w in PDL_PROOF_WORDS.items()}
    nn, pr = _struct(N_.PdlNonces, nonces), _struct(N_.PdlProof, out)
    N_.check(N_.lib.mpe_pdl_prove(ctx.h, pk.h, stm.h, B, _ptr(d_key_idx), _ptr(d_st_idx), _ptr(d_cipher), _ptr(d_Q), _ptr(d_G),
                                  _ptr(d_x), _ptr(d_r), C.byref(nn), C.byref(pr), ctx.stream()), "mpe_pdl_prove")
    return out


def pdl_verify(ctx, pk, stm, d_cipher, d_Q, d_G, proof, d_key_idx=None, d_st_idx=None):
    B = d_cipher.shape[0]
    ok = _flags(ctx, B)
    pr = _struct(N_.PdlProof, proof)
    N_.check(N_.lib.mpe_pdl_verify(ctx.h, pk.h, stm.h, B, _ptr(d_key_idx), _ptr(d_st_idx), _ptr(d_cipher), _ptr(d_Q), _ptr(d_G),
                                   C.byref(pr), _ptr(ok), ctx.stream()), "mpe_pdl_verify")
    return ok


# ================================================================================================
# GG20 signing: key object, the per-party round view, the lock-step composition
# ================================================================================================
GG20_ROUNDS = [0, 1, 2, 3, 4, 5, 7]             # rounds that emit a message


def gg20_msg_words(S, n, rnd):
    return N_.lib.mpe_gg20_msg_words(S, n, rnd)


class Gg20Keys:
    """`LocalKey` material (keygen/rounds.rs:311-322) + the signer set, resident in HBM.
    arrays: dict of numpy uint32 arrays for ALL n parties of every key set: x [K*n,8], p, q [K*n,32], Nt, h1, h2 [K*n,64],
    y [K,16], X [K*n,16] (and optionally N [K*n,64], else p*q).  own: the party indices whose SECRETS (x, p, q) this
    object gets (default: all — the Simulation harness); the other parties' rows of x, p, q never leave the host."""

    def __init__(self, ctx, t, n, signers, arrays, own=None, nkeysets=1):
        self.ctx, self.t, self.n, self.S, self.K = ctx, t, n, len(signers), nkeysets
        own = list(range(n)) if own is None else sorted(int(a) for a in own)
        self.own = own
        a = {f: np.ascontiguousarray(arrays[f]) for f in ("x", "p", "q", "Nt", "h1", "h2", "y", "X")}
        if "N" in arrays and arrays["N"] is not None:
            a["N"] = np.ascontiguousarray(arrays["N"])
        else:
            ps, qs = words_to_ints(a["p"]), words_to_ints(a["q"])
            a["N"] = ints_to_words([p_ * q_ for p_, q_ in zip(ps, qs)], 64)
        rows = [kk * n + o for kk in range(nkeysets) for o in own]
        h = {f: a[f] for f in ("N", "Nt", "h1", "h2", "y", "X")}
        for f in ("x", "p", "q"):
            h[f] = np.ascontiguousarray(a[f][rows])
        self.d = {f: torch.from_numpy(h[f].view(np.int32)).to(ctx.device) for f in h}
        sg = (C.c_int32 * len(signers))(*[int(s) for s in signers])
        ow = (C.c_int32 * len(own))(*own)
        hd = C.c_void_p()
        N_.check(N_.lib.mpe_gg20_keys_create(ctx.h, t, n, len(signers), sg, nkeysets, len(own), ow, *[_ptr(self.d[f]) for f in
                                             ("x", "p", "q", "N", "Nt", "h1", "h2", "y", "X")], C.byref(hd), ctx.stream()),
                 "mpe_gg20_keys_create")
        self.h = hd

    def fb_window_bits(self):
        return N_.lib.mpe_gg20_keys_fb_window_bits(self.h)

    def close(self):
        if self.h:
            N_.lib.mpe_gg20_keys_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Gg20Session:
    """`RoundN::proceed` for B sessions x the local parties `local` (signer ordinals).  nonces: dict of device int32
    tensors with leading dimensions [B][len(local)] (fields _native.GG20_NONCE_FIELDS; `msg` optional)."""

    def __init__(self, ctx, keys, B, local, nonces, keyset=None, dedup_verify=False):
        self.ctx, self.keys, self.B, self.local = ctx, keys, B, [int(x) for x in local]
        self.S, self.n, self.L = keys.S, keys.n, len(local)
        self._nonces = dict(nonces)
        if "msg" not in self._nonces:
            self._nonces["msg"] = torch.zeros((B, 8), dtype=torch.int32, device=ctx.device)
        self._keyset = keyset
        nn = _struct(N_.Gg20Nonces, self._nonces)
        lc = (C.c_int32 * self.L)(*self.local)
        h = C.c_void_p()
        N_.check(N_.lib.mpe_gg20_session_create(ctx.h, keys.h, B, self.L, lc, _ptr(keyset), C.byref(nn), int(bool(dedup_verify)),
                                                C.byref(h), ctx.stream()), "mpe_gg20_session_create")
        self.h = h

    def _off(self, in_off):
        return None if in_off is None else (C.c_int64 * self.S)(*[int(x) for x in in_off])

    def rearm(self, nonces, keyset=None):
        """the next batch of the same shape on this object (mpe_gg20_session_rearm): fresh sampled values, rounds from 0"""
        self._nonces = dict(nonces)
        if "msg" not in self._nonces:
            self._nonces["msg"] = torch.zeros((self.B, 8), dtype=torch.int32, device=self.ctx.device)
        self._keyset = keyset
        nn = _struct(N_.Gg20Nonces, self._nonces)
        N_.check(N_.lib.mpe_gg20_session_rearm(self.h, _ptr(keyset), C.byref(nn), self.ctx.stream()), "mpe_gg20_session_rearm")

    def abort(self):
        """mpe_gg20_session_abort: give the running batch up (state wiped now); rearm() starts the next one"""
        N_.check(N_.lib.mpe_gg20_session_abort(self.h, self.ctx.stream()), "mpe_gg20_session_abort")

    def round(self, rnd, d_in=None, in_off=None, msg=None, out=None):
        """Runs round `rnd` (0..7, 8 = SignManual::complete).  d_in: the previous round's records of all S senders (device
        int32 tensor; sender j's [B][W] block at record in_off[j], default j*B).  Returns this object's outgoing records
        [L, B, W] (None for rounds 6 and 8); `out`: a contiguous [L, B, W] int32 device tensor to write them into (e.g. this
        rank's slot of an all-gather buffer)."""
        lib, st = N_.lib, self.ctx.stream()
        W = gg20_msg_words(self.S, self.n, rnd) if rnd in GG20_ROUNDS else 0
        if out is not None and W:
            if not (out.is_contiguous() and out.dtype == torch.int32 and out.numel() == self.L * self.B * W and out.device == self.ctx.device):
                raise ValueError("round(out=...): need a contiguous int32 [L, B, W] tensor on the context's device")
        else:
            out = torch.empty((self.L, self.B, W), dtype=torch.int32, device=self.ctx.device) if W else None
        if rnd == 0:
            N_.check(lib.mpe_gg20_round0(self.h, _ptr(out), st), "mpe_gg20_round0")
        elif 1 <= rnd <= 5:
            N_.check(getattr(lib, f"mpe_gg20_round{rnd}")(self.h, _ptr(d_in), self._off(in_off), _ptr(out), st), f"mpe_gg20_round{rnd}")
        elif rnd == 6:
            N_.check(lib.mpe_gg20_round6(self.h, _ptr(d_in), self._off(in_off), st), "mpe_gg20_round6")
        elif rnd == 7:
            N_.check(lib.mpe_gg20_round7(self.h, _ptr(msg), _ptr(out), st), "mpe_gg20_round7")
        elif rnd == 8:
            N_.check(lib.mpe_gg20_complete(self.h, _ptr(d_in), self._off(in_off), st), "mpe_gg20_complete")
        else:
            raise ValueError(rnd)
        return out

    def fault_inject(self, step, party_mask):
        N_.check(N_.lib.mpe_gg20_session_fault_inject(self.h, int(step), int(party_mask)), "mpe_gg20_session_fault_inject")

    def blame6_state(self, d_nonce):
        """(miu [L,B,S-1,64], a1, a2 [L,B,16], z [L,B,8]): what the local parties publish for the phase-6 blame"""
        L, B, S, dv = self.L, self.B, self.S, self.ctx.device
        miu = torch.empty((L, B, S - 1, 64), dtype=torch.int32, device=dv)
        a1, a2, z = torch.empty((L, B, 16), dtype=torch.int32, device=dv), torch.empty((L, B, 16), dtype=torch.int32, device=dv), torch.empty((L, B, 8), dtype=torch.int32, device=dv)
        N_.check(N_.lib.mpe_gg20_session_blame6_state(self.h, _ptr(d_nonce), _ptr(miu), _ptr(a1), _ptr(a2), _ptr(z), self.ctx.stream()),
                 "mpe_gg20_session_blame6_state")
        return miu, a1, a2, z

    def result(self, signature=True):
        """dict of device tensors: status, bad_actors, recid [L,B]; r, s [L,B,8]; R [L,B,16]"""
        L, B, dv = self.L, self.B, self.ctx.device
        o = dict(status=torch.empty((L, B), dtype=torch.int32, device=dv), bad_actors=torch.empty((L, B), dtype=torch.int32, device=dv),
                 R=torch.empty((L, B, 16), dtype=torch.int32, device=dv))
        if signature:
            o.update(r=torch.empty((L, B, 8), dtype=torch.int32, device=dv), s=torch.empty((L, B, 8), dtype=torch.int32, device=dv),
                     recid=torch.empty((L, B), dtype=torch.int32, device=dv))
        N_.check(N_.lib.mpe_gg20_session_result(self.h, _ptr(o["status"]), _ptr(o["bad_actors"]), _ptr(o.get("r")), _ptr(o.get("s")),
                                                _ptr(o.get("recid")), _ptr(o["R"]), self.ctx.stream()), "mpe_gg20_session_result")
        return o

    def close(self):
        if self.h:
            N_.lib.mpe_gg20_session_destroy(self.h, self.ctx.stream())
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gg20_sign(ctx, keys, nonces, B, dedup_verify=False, chunk=0, want_R=False, keyset=None):
    """nonces: dict of device int32 tensors (fields _native.GG20_NONCE_FIELDS, [B][S] layout).  Returns device tensors
    r [B,8], s [B,8], recid [B], status [B] (0 = signed and verified) and optionally R [B,16]."""
    r, s = _new(ctx, B, 8), _new(ctx, B, 8)
    recid = torch.empty((B,), dtype=torch.int32, device=ctx.device)
    status = torch.full((B,), -1, dtype=torch.int32, device=ctx.device)
    R = _new(ctx, B, 16) if want_R else None
    nn = _struct(N_.Gg20Nonces, nonces)
    N_.check(N_.lib.mpe_gg20_sign(ctx.h, keys.h, B, _ptr(keyset), C.byref(nn), _ptr(r), _ptr(s), _ptr(recid), _ptr(R), _ptr(status),
                                  int(bool(dedup_verify)), int(chunk), ctx.stream()), "mpe_gg20_sign")
    return (r, s, recid, status, R) if want_R else (r, s, recid, status)


PLACE_PARTY, PLACE_ROTATED = 0, 1


class Comm:
    """mpe_comm_*: the RCCL communicator of the round fan-out behind the C-ABI (include/mpecdsa_hip.h).  `exchange_id(id_or_None)`:
    a callable that carries rank 0's 128 id bytes to every rank (e.g. a torch.distributed broadcast, a file, a socket)."""
    ID_BYTES = 128

    def __init__(self, ctx, rank, world, exchange_id=None):
        self.ctx = ctx
        buf = C.create_string_buffer(self.ID_BYTES)
        if rank == 0:
            N_.check(N_.lib.mpe_comm_unique_id(buf), "mpe_comm_unique_id")
        ident = buf.raw
        if world > 1 or exchange_id is not None:
            ident = bytes(exchange_id(ident if rank == 0 else None))
        h = C.c_void_p()
        N_.check(N_.lib.mpe_comm_create(ctx.h, ident, rank, world, C.byref(h)), "mpe_comm_create")
        self.h, self.rank, self.world = h, rank, world

    def layout_self_test(self, rows_per_rank):
        mode, ok = C.c_int(-1), C.c_int(0)
        N_.check(N_.lib.mpe_comm_layout_self_test(self.h, rows_per_rank, C.byref(mode), C.byref(ok), self.ctx.stream()), "mpe_comm_layout_self_test")
        return dict(mode=("inplace", "copy")[mode.value], ok=bool(ok.value))

    def all_gather(self, buf, bytes_per_rank):
        N_.check(N_.lib.mpe_comm_all_gather(self.h, _ptr(buf), bytes_per_rank, self.ctx.stream()), "mpe_comm_all_gather")

    def round_exchange(self, S, n, rnd, per_rank, batch, slab):
        N_.check(N_.lib.mpe_gg20_round_exchange(self.h, S, n, rnd, per_rank, batch, _ptr(slab), self.ctx.stream()), "mpe_gg20_round_exchange")

    def close(self):
        if self.h:
            N_.lib.mpe_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_library():
    """mpe_comm_library: {"path": the librccl the communicator entry points are bound to, "adopted": it was already in the process
    (PyTorch's own copy), "version": ncclGetVersion}"""
    buf = C.create_string_buffer(4096)
    ad, ver = C.c_int(0), C.c_int(0)
    N_.check(N_.lib.mpe_comm_library(buf, 4096, C.byref(ad), C.byref(ver)), "mpe_comm_library")
    return dict(path=buf.value.decode(), adopted=bool(ad.value), version=ver.value)


def shard_where(placement, S, world, block, party):
    r, s = C.c_int(0), C.c_int(0)
    N_.check(N_.lib.mpe_gg20_shard_where(placement, S, world, block, party, C.byref(r), C.byref(s)), "mpe_gg20_shard_where")
    return r.value, s.value


def shard_in_off(placement, S, world, batch, block):
    off = (C.c_int64 * S)()
    N_.check(N_.lib.mpe_gg20_shard_in_off(placement, S, world, batch, block, off), "mpe_gg20_shard_in_off")
    return list(off)


class Gg20Pipeline:
    """mpe_gg20_pipeline_*: a stream of `batch`-session batches, `group` of them coalesced per pass, `lanes` passes in flight on one
    stream each (include/mpecdsa_hip.h).  submit / submit_seeded return a ticket; the result tensors of a ticket are valid after
    wait(ticket) (or after stream_wait on the consuming stream)."""

    def __init__(self, ctx, keys, batch, group=4, lanes=2, dedup_verify=False):
        self.ctx, self.keys, self.batch, self.group, self.lanes = ctx, keys, batch, group, lanes
        h = C.c_void_p()
        N_.check(N_.lib.mpe_gg20_pipeline_create(ctx.h, keys.h, batch, group, lanes, int(bool(dedup_verify)), C.byref(h)), "mpe_gg20_pipeline_create")
        self.h = h
        self._keep = {}                       # ticket -> the tensors the device still reads / writes

    def _outs(self, want_R):
        B = self.batch
        r, s = _new(self.ctx, B, 8), _new(self.ctx, B, 8)
        recid = torch.empty((B,), dtype=torch.int32, device=self.ctx.device)
        status = torch.full((B,), -1, dtype=torch.int32, device=self.ctx.device)
        R = _new(self.ctx, B, 16) if want_R else None
        return r, s, recid, status, R

    def submit(self, nonces, keyset=None, want_R=False):
        r, s, recid, status, R = self._outs(want_R)
        nn = _struct(N_.Gg20Nonces, nonces)
        t = C.c_uint64(0)
        N_.check(N_.lib.mpe_gg20_pipeline_submit(self.h, _ptr(keyset), C.byref(nn), _ptr(r), _ptr(s), _ptr(recid), _ptr(R), _ptr(status), self.ctx.stream(),
                                                 C.byref(t)), "mpe_gg20_pipeline_submit")
        self._keep[t.value] = (nonces, keyset, r, s, recid, status, R)
        return t.value

    def submit_seeded(self, seed, batch_counter, msg, keyset=None, want_R=False):
        r, s, recid, status, R = self._outs(want_R)
        t = C.c_uint64(0)
        N_.check(N_.lib.mpe_gg20_pipeline_submit_seeded(self.h, _ptr(keyset), _seed(seed), int(batch_counter), _ptr(msg), _ptr(r), _ptr(s), _ptr(recid),
                                                        _ptr(R), _ptr(status), self.ctx.stream(), C.byref(t)), "mpe_gg20_pipeline_submit_seeded")
        self._keep[t.value] = (msg, keyset, r, s, recid, status, R)
        return t.value

    def flush(self):
        """sends the open group; a pass that fails is reported by its tickets (wait / ticket_rc), not here"""
        N_.lib.mpe_gg20_pipeline_flush(self.h)

    def done(self, ticket):
        """True once the batch is complete — also when its pass FAILED (ticket_rc / wait tell)"""
        d = C.c_int(0)
        N_.lib.mpe_gg20_pipeline_query(self.h, ticket, C.byref(d))
        return bool(d.value)

    def ticket_rc(self, ticket):
        """(launched, rc) of the pass that carries the batch (rc = MPE_OK while its group is still open)"""
        la, rc = C.c_int(0), C.c_int(0)
        N_.check(N_.lib.mpe_gg20_pipeline_ticket_rc(self.h, ticket, C.byref(la), C.byref(rc)), "mpe_gg20_pipeline_ticket_rc")
        return bool(la.value), rc.value

    def wait(self, ticket, want_R=False, check=True):
        """blocks until the batch is complete; returns (r, s, recid, status[, R]) and forgets the ticket's tensors.  A batch whose PASS
        failed raises MpeError (check=True) or returns its arrays — status = MPE_GG20_STATUS_PASS_FAILED(rc), no signature — with
        check=False"""
        rc = N_.lib.mpe_gg20_pipeline_wait(self.h, ticket)
        if ticket not in self._keep:
            N_.check(rc if rc != N_.MPE_OK else N_.MPE_E_ARG, "mpe_gg20_pipeline_wait (unknown ticket)")
        _, _, r, s, recid, status, R = self._keep.pop(ticket)
        if check:
            N_.check(rc, "mpe_gg20_pipeline_wait")
        return (r, s, recid, status, R) if want_R else (r, s, recid, status)

    def set_deadline_us(self, us):
        N_.check(N_.lib.mpe_gg20_pipeline_set_deadline_us(self.h, int(us)), "mpe_gg20_pipeline_set_deadline_us")

    def set_eager(self, on=True):
        N_.check(N_.lib.mpe_gg20_pipeline_set_eager(self.h, int(bool(on))), "mpe_gg20_pipeline_set_eager")

    def poll(self):
        la = C.c_int(0)
        N_.check(N_.lib.mpe_gg20_pipeline_poll(self.h, C.byref(la)), "mpe_gg20_pipeline_poll")
        return bool(la.value)

    def inject_fault(self, passes=1, rc=N_.MPE_E_NOMEM):
        N_.check(N_.lib.mpe_gg20_pipeline_inject_fault(self.h, int(passes), int(rc)), "mpe_gg20_pipeline_inject_fault")

    def counters(self):
        v = [C.c_uint64(0) for _ in range(4)]
        N_.check(N_.lib.mpe_gg20_pipeline_counters(self.h, *[C.byref(x) for x in v]), "mpe_gg20_pipeline_counters")
        return dict(zip(("groups", "by_deadline", "by_idle", "failed"), (x.value for x in v)))

    def latency_ms(self, ticket):
        ms = C.c_float(0)
        N_.check(N_.lib.mpe_gg20_pipeline_latency_ms(self.h, ticket, C.byref(ms)), "mpe_gg20_pipeline_latency_ms")
        return ms.value

    def pass_ms(self, ticket):
        ms = C.c_float(0)
        N_.check(N_.lib.mpe_gg20_pipeline_pass_ms(self.h, ticket, C.byref(ms)), "mpe_gg20_pipeline_pass_ms")
        return ms.value

    def sampler_failures(self):
        v = C.c_int32(0)
        N_.check(N_.lib.mpe_gg20_pipeline_sampler_failures(self.h, C.byref(v)), "mpe_gg20_pipeline_sampler_failures")
        return v.value

    def close(self):
        if self.h:
            N_.lib.mpe_gg20_pipeline_destroy(self.h)
            self.h = None
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the sampling side of the trait surface (mpe_sample.h): curv Samplable / from_modulo / Scalar::random on the device ----
SAMPLE_NONZERO, SAMPLE_PLUS_ONE, SAMPLE_COPRIME = 1, 2, 4


def _seed(seed):
    b = bytes(seed)
    if len(b) != 32:
        raise ValueError("the sampler's seed is 32 bytes")
    return b


def sample_bits(ctx, batch, seed, stream_id, bits, out_words):
    out = _new(ctx, batch, out_words)
    N_.check(N_.lib.mpe_sample_bits(ctx.h, batch, _seed(seed), stream_id, bits, out_words, _ptr(out), ctx.stream()), "mpe_sample_bits")
    return out


def sample_below(ctx, batch, seed, stream_id, d_bound, out_words, d_bound_idx=None, flags=0):
    """d_bound: device int32 [nbounds, bound_words]; returns (values [batch, out_words], failures as a device int32 [1])"""
    out = _new(ctx, batch, out_words)
    fail = torch.zeros((1,), dtype=torch.int32, device=ctx.device)
    N_.check(N_.lib.mpe_sample_below(ctx.h, batch, _seed(seed), stream_id, _ptr(d_bound), d_bound.shape[1], d_bound.shape[0], _ptr(d_bound_idx), flags,
                                     out_words, _ptr(out), _ptr(fail), ctx.stream()), "mpe_sample_below")
    return out, fail


def sample_scalar(ctx, batch, seed, stream_id):
    out = _new(ctx, batch, 8)
    fail = torch.zeros((1,), dtype=torch.int32, device=ctx.device)
    N_.check(N_.lib.mpe_sample_scalar(ctx.h, batch, _seed(seed), stream_id, _ptr(out), _ptr(fail), ctx.stream()), "mpe_sample_scalar")
    return out, fail


def gg20_nonce_shapes(S, n, L, B):
    """rows x words of every field of mpe_gg20_nonces (include/mpecdsa_hip.h) for B sessions x L local parties"""
    P = L * (S - 1)
    return dict(k=(B * L, 8), gamma=(B * L, 8), blind=(B * L, 8), r_a=(B * L, 64), al_alpha=(B * L * n, 24), al_beta=(B * L * n, 64),
                al_gamma=(B * L * n, 88), al_rho=(B * L * n, 72), mb_beta_tag=(B * P * 2, 64), mb_r=(B * P * 2, 64), mb_nonce_b=(B * P * 2, 8),
                mb_nonce_bt=(B * P * 2, 8), l=(B * L, 8), ped_s1=(B * L, 8), ped_s2=(B * L, 8), pdl_alpha=(B * P, 24), pdl_beta=(B * P, 64),
                pdl_rho=(B * P, 72), pdl_gamma=(B * P, 88), heg_s1=(B * L, 8), heg_s2=(B * L, 8), msg=(B, 8))


def gg20_sample_nonces(ctx, keys, B, seed, batch_counter, local=None, keyset=None, msg=None, out=None):
    """mpe_gg20_sample_nonces: every value the local parties of B sessions draw while signing, from (seed, batch_counter).
    Returns (dict of device tensors in the layout gg20_sign / Gg20Session take, failures [1]); `msg` (device [B, 8]) is passed through."""
    local = list(range(keys.S)) if local is None else list(local)
    if out is None:
        out = {f: torch.zeros(shape, dtype=torch.int32, device=ctx.device) for f, shape in gg20_nonce_shapes(keys.S, keys.n, len(local), B).items()}
    if msg is not None:
        out["msg"] = msg
    fail = torch.zeros((1,), dtype=torch.int32, device=ctx.device)
    nn = _struct(N_.Gg20Nonces, out)
    lc = (C.c_int32 * len(local))(*local)
    N_.check(N_.lib.mpe_gg20_sample_nonces(ctx.h, keys.h, B, len(local), lc, _ptr(keyset), _seed(seed), int(batch_counter), C.byref(nn), _ptr(fail),
                                           ctx.stream()), "mpe_gg20_sample_nonces")
    return out, fail


# ---- keygen verification math (gg_2020/party_i.rs:260-438) ----
def correct_key_verify(ctx, d_N, d_sigma):
    ok = _flags(ctx, d_N.shape[0])
    N_.check(N_.lib.mpe_correct_key_verify(ctx.h, d_N.shape[0], _ptr(d_N), _ptr(d_sigma), _ptr(ok), ctx.stream()), "mpe_correct_key_verify")
    return ok


def composite_dlog_verify(ctx, d_N, d_g, d_ni, d_x, d_y):
    ok = _flags(ctx, d_N.shape[0])
    N_.check(N_.lib.mpe_composite_dlog_verify(ctx.h, d_N.shape[0], _ptr(d_N), _ptr(d_g), _ptr(d_ni), _ptr(d_x), _ptr(d_y), _ptr(ok),
                                              ctx.stream()), "mpe_composite_dlog_verify")
    return ok


def keygen_verify_round1(ctx, n_parties, msgs):
    """`phase1_verify_com_phase3_verify_correct_key_verify_dlog_phase2_distribute` (party_i.rs:260-320) over items = (session, prover):
    msgs = dict of device tensors (fields _native.KEYGEN_ROUND1_FIELDS).  Returns (ok [B] uint8, bad_actors [B / n_parties] int32 masks)."""
    B = msgs["N"].shape[0]
    ok = _flags(ctx, B)
    bad = torch.zeros((B // n_parties,), dtype=torch.int32, device=ctx.device)
    st = _struct(N_.KeygenRound1, msgs)
    N_.check(N_.lib.mpe_keygen_verify_round1(ctx.h, B, n_parties, C.byref(st), _ptr(ok), _ptr(bad), ctx.stream()), "mpe_keygen_verify_round1")
    return ok, bad


def keygen_verify_round2(ctx, n_parties, t1, d_commits, d_share, d_index, d_y):
    """the verdict of `phase2_verify_vss_construct_keypair_phase3_pok_dlog` (party_i.rs:322-367)"""
    B = d_share.shape[0]
    ok = _flags(ctx, B)
    bad = torch.zeros((B // n_parties,), dtype=torch.int32, device=ctx.device)
    N_.check(N_.lib.mpe_keygen_verify_round2(ctx.h, B, n_parties, t1, _ptr(d_commits), _ptr(d_share), _ptr(d_index), _ptr(d_y), _ptr(ok), _ptr(bad),
                                             ctx.stream()), "mpe_keygen_verify_round2")
    return ok, bad


def correct_key_prove(ctx, sk):
    """`NiCorrectKeyProof::proof` for every key of a private key set: sigma [nkeys, 11, 64]"""
    sigma = torch.zeros((sk.nkeys, 11, 64), dtype=torch.int32, device=ctx.device)
    N_.check(N_.lib.mpe_correct_key_prove(ctx.h, sk.h, _ptr(sigma), ctx.stream()), "mpe_correct_key_prove")
    return sigma


def composite_dlog_prove(ctx, d_N, d_g, d_ni, d_secret, d_r):
    B = d_N.shape[0]
    x, y = _new(ctx, B, 64), _new(ctx, B, 73)
    N_.check(N_.lib.mpe_composite_dlog_prove(ctx.h, B, _ptr(d_N), _ptr(d_g), _ptr(d_ni), _ptr(d_secret), _ptr(d_r), _ptr(x), _ptr(y), ctx.stream()),
             "mpe_composite_dlog_prove")
    return x, y


def vss_validate_share(ctx, t1, d_commits, d_share, d_index):
    ok = _flags(ctx, d_share.shape[0])
    N_.check(N_.lib.mpe_vss_validate_share(ctx.h, d_share.shape[0], t1, _ptr(d_commits), _ptr(d_share), _ptr(d_index), _ptr(ok), ctx.stream()),
             "mpe_vss_validate_share")
    return ok


def vss_point_commitment(ctx, t1, d_commits, d_index):
    out = _new(ctx, d_index.shape[0], 16)
    N_.check(N_.lib.mpe_vss_point_commitment(ctx.h, d_index.shape[0], t1, _ptr(d_commits), _ptr(d_index), _ptr(out), ctx.stream()),
             "mpe_vss_point_commitment")
    return out


# ---- identifiable abort (gg_2020/blame.rs) ----
def gg20_blame5(ctx, keys, B, opened, keyset=None):
    """opened: dict of device tensors (fields _native.Blame5In) -> bad_actors bit masks [B] (device int32)"""
    bad = torch.empty((B,), dtype=torch.int32, device=ctx.device)
    st_ = _struct(N_.Blame5In, opened)
    N_.check(N_.lib.mpe_gg20_blame5(ctx.h, keys.h, B, _ptr(keyset), C.byref(st_), _ptr(bad), ctx.stream()), "mpe_gg20_blame5")
    return bad


def gg20_blame6(ctx, keys, B, opened, keyset=None):
    bad = torch.empty((B,), dtype=torch.int32, device=ctx.device)
    st_ = _struct(N_.Blame6In, opened)
    N_.check(N_.lib.mpe_gg20_blame6(ctx.h, keys.h, B, _ptr(keyset), C.byref(st_), _ptr(bad), ctx.stream()), "mpe_gg20_blame6")
    return bad


def gg20_blame7(ctx, S, B, opened):
    bad = torch.empty((B,), dtype=torch.int32, device=ctx.device)
    st_ = _struct(N_.Blame7In, opened)
    N_.check(N_.lib.mpe_gg20_blame7(ctx.h, S, B, C.byref(st_), _ptr(bad), ctx.stream()), "mpe_gg20_blame7")
    return bad


def ecddh_prove(ctx, d_x, d_s, statement):
    B = d_x.shape[0]
    out = dict(a1=_new(ctx, B, 16), a2=_new(ctx, B, 16), z=_new(ctx, B, 8))
    stt, pr = _struct(N_.EcddhStatement, statement), _struct(N_.EcddhProof, out)
    N_.check(N_.lib.mpe_ecddh_prove(ctx.h, B, _ptr(d_x), _ptr(d_s), C.byref(stt), C.byref(pr), ctx.stream()), "mpe_ecddh_prove")
    return out


def ecddh_verify(ctx, statement, proof):
    B = proof["z"].shape[0]
    ok = _flags(ctx, B)
    stt, pr = _struct(N_.EcddhStatement, statement), _struct(N_.EcddhProof, proof)
    N_.check(N_.lib.mpe_ecddh_verify(ctx.h, B, C.byref(stt), C.byref(pr), _ptr(ok), ctx.stream()), "mpe_ecddh_verify")
    return ok


# ---- curv sigma proofs of phases 3 / 6, the phase-1 commitment (device-tensor API) ----
def pedersen_prove(ctx, d_m, d_r, d_s1, d_s2):
    B = d_m.shape[0]
    out = dict(com=_new(ctx, B, 16), e=_new(ctx, B, 8), a1=_new(ctx, B, 16), a2=_new(ctx, B, 16), z1=_new(ctx, B, 8), z2=_new(ctx, B, 8))
    pr = _struct(N_.PedersenProof, out)
    N_.check(N_.lib.mpe_pedersen_prove(ctx.h, B, _ptr(d_m), _ptr(d_r), _ptr(d_s1), _ptr(d_s2), C.byref(pr), ctx.stream()), "mpe_pedersen_prove")
    return out


def pedersen_verify(ctx, proof):
    B = proof["com"].shape[0]
    ok = _flags(ctx, B)
    pr = _struct(N_.PedersenProof, proof)
    N_.check(N_.lib.mpe_pedersen_verify(ctx.h, B, C.byref(pr), _ptr(ok), ctx.stream()), "mpe_pedersen_verify")
    return ok


def heg_prove(ctx, d_x, d_r, d_s1, d_s2, statement):
    B = d_x.shape[0]
    out = dict(T=_new(ctx, B, 16), A3=_new(ctx, B, 16), z1=_new(ctx, B, 8), z2=_new(ctx, B, 8))
    stt, pr = _struct(N_.HegStatement, statement), _struct(N_.HegProof, out)
    N_.check(N_.lib.mpe_heg_prove(ctx.h, B, _ptr(d_x), _ptr(d_r), _ptr(d_s1), _ptr(d_s2), C.byref(stt), C.byref(pr), ctx.stream()), "mpe_heg_prove")
    return out


def heg_verify(ctx, statement, proof):
    B = proof["T"].shape[0]
    ok = _flags(ctx, B)
    stt, pr = _struct(N_.HegStatement, statement), _struct(N_.HegProof, proof)
    N_.check(N_.lib.mpe_heg_verify(ctx.h, B, C.byref(stt), C.byref(pr), _ptr(ok), ctx.stream()), "mpe_heg_verify")
    return ok


def hash_commit_point(ctx, d_P, d_blind):
    out = _new(ctx, d_P.shape[0], 8)
    N_.check(N_.lib.mpe_hash_commit_point(ctx.h, d_P.shape[0], _ptr(d_P), _ptr(d_blind), _ptr(out), ctx.stream()), "mpe_hash_commit_point")
    return out


BOB_PROOF_WORDS = dict(t=64, z=64, e=8, s=64, s1=25, s2=89, t1=81, t2=89)
BOB_NONCE_WORDS = dict(alpha=24, beta=64, gamma=80, rho=72, rho_prim=88, sigma=72, tau=88)


def bob_generate(ctx, pk, stm, d_a_enc, d_mta_enc, d_b, d_beta_prim, d_r, nonces, check, d_key_idx=None, d_st_idx=None):
    """`BobProof::generate(a_encrypted, mta_encrypted, b, beta_prim, alice_ek, dlog_statement, r, check)` batched.
    Returns (proof dict, u or None)."""
    B = d_b.shape[0]
    out = {f: _new(ctx, B, w) for f, w in BOB_PROOF_WORDS.items()}
    u = _new(ctx, B, 16) if check else None
    nn, pr = _struct(N_.BobNonces, nonces), _struct(N_.BobProof, out)
    N_.check(N_.lib.mpe_bob_generate(ctx.h, pk.h, stm.h, B, _ptr(d_key_idx), _ptr(d_st_idx), _ptr(d_a_enc), _ptr(d_mta_enc),
                                     _ptr(d_b), _ptr(d_beta_prim), _ptr(d_r), C.byref(nn), int(bool(check)), C.byref(pr),
                                     _ptr(u), ctx.stream()), "mpe_bob_generate")
    return out, u


def bob_verify(ctx, pk, stm, d_a_enc, d_mta_enc, proof, d_X=None, d_u=None, d_key_idx=None, d_st_idx=None):
    """`BobProof::verify` (d_X, d_u None) / `BobProofExt::verify` batched -> ok flags"""
    B = d_a_enc.shape[0]
    ok = _flags(ctx, B)
    pr = _struct(N_.BobProof, proof)
    N_.check(N_.lib.mpe_bob_verify(ctx.h, pk.h, stm.h, B, _ptr(d_key_idx), _ptr(d_st_idx), _ptr(d_a_enc), _ptr(d_mta_enc),
                                   C.byref(pr), _ptr(d_X), _ptr(d_u), _ptr(ok), ctx.stream()), "mpe_bob_verify")
    return ok


# ================================================================================================
# MtA (src/utilities/mta/mod.rs): MessageA / MessageB / verify_proofs_get_alpha, batched
# ================================================================================================
def mta_message_a(ctx, pk, stm, d_a, d_r, nonces, d_key_idx=None):
    """`MessageA::a_with_predefined_randomness`: returns (c [B,128], range proofs dict [B*nst, ...])"""
    B, total = d_a.shape[0], d_a.shape[0] * stm.count
    c = _new(ctx, B, 128)
    proofs = {f: _new(ctx, total, w) for f, w in ALICE_PROOF_WORDS.items()}
    nn, pr = _struct(N_.AliceNonces, nonces), _struct(N_.AliceProof, proofs)
    N_.check(N_.lib.mpe_mta_message_a(ctx.h, pk.h, stm.h, B, _ptr(d_key_idx), _ptr(d_a), _ptr(d_r), C.byref(nn), _ptr(c),
                                      C.byref(pr), ctx.stream()), "mpe_mta_message_a")
    return c, proofs


def mta_message_b(ctx, pk, stm, d_b, d_ca, range_proofs, d_r, d_beta_tag, d_nonce_b, d_nonce_bt, d_key_idx=None):
    """`MessageB::b_with_predefined_randomness`: returns dict(c, beta, b_proof, beta_tag_proof, ok)"""
    B = d_b.shape[0]
    out = dict(c=_new(ctx, B, 128), beta=_new(ctx, B, 8), ok=_flags(ctx, B),
               b_proof=dict(pk=_new(ctx, B, 16), R=_new(ctx, B, 16), z=_new(ctx, B, 8)),
               beta_tag_proof=dict(pk=_new(ctx, B, 16), R=_new(ctx, B, 16), z=_new(ctx, B, 8)))
    rp = _struct(N_.AliceProof, range_proofs)
    p1, p2 = _struct(N_.DlogProof, out["b_proof"]), _struct(N_.DlogProof, out["beta_tag_proof"])
    N_.check(N_.lib.mpe_mta_message_b(ctx.h, pk.h, stm.h, B, _ptr(d_key_idx), _ptr(d_b), _ptr(d_ca), C.byref(rp), _ptr(d_r),
                                      _ptr(d_beta_tag), _ptr(d_nonce_b), _ptr(d_nonce_bt), _ptr(out["c"]), _ptr(out["beta"]),
                                      C.byref(p1), C.byref(p2), _ptr(out["ok"]), ctx.stream()), "mpe_mta_message_b")
    return out


def mta_verify_get_alpha(ctx, sk, d_cb, b_proof, beta_tag_proof, d_a, d_key_idx=None):
    """`MessageB::verify_proofs_get_alpha(dk, a)`: returns (alpha [B,8], alice_share [B,64], ok)"""
    B = d_cb.shape[0]
    alpha, share, ok = _new(ctx, B, 8), _new(ctx, B, 64), _flags(ctx, B)
    p1, p2 = _struct(N_.DlogProof, b_proof), _struct(N_.DlogProof, beta_tag_proof)
    N_.check(N_.lib.mpe_mta_verify_get_alpha(ctx.h, sk.h, B, _ptr(d_key_idx), _ptr(d_cb), C.byref(p1), C.byref(p2), _ptr(d_a),
                                             _ptr(alpha), _ptr(share), _ptr(ok), ctx.stream()), "mpe_mta_verify_get_alpha")
    return alpha, share, ok


# ---- Lindell'17 two-party ECDSA, signing (lindell_2017/party_two.rs:390-423, party_one.rs:519-565) ----
def lindell_partial_sig(ctx, pk, d_c_key, d_x2, d_k2, d_R1, d_msg, d_rho, d_r, d_key_idx=None):
    """`PartialSig::compute` batched: returns c3 [B,128] (device)."""
    B = d_c_key.shape[0]
    c3 = _new(ctx, B, 128)
    N_.check(N_.lib.mpe_lindell_partial_sig(ctx.h, pk.h, B, _ptr(d_key_idx), _ptr(d_c_key), _ptr(d_x2), _ptr(d_k2), _ptr(d_R1),
                                            _ptr(d_msg), _ptr(d_rho), _ptr(d_r), _ptr(c3), ctx.stream()), "mpe_lindell_partial_sig")
    return c3


def paillier_open(ctx, sk, d_c, d_key_idx=None):
    """kzen-paillier `Open::open`: returns (m [B,64], r [B,64]) with c = (1 + m N) r^N mod N^2."""
    B = d_c.shape[0]
    m, r = _new(ctx, B, 64), _new(ctx, B, 64)
    N_.check(N_.lib.mpe_paillier_open(ctx.h, sk.h, B, _ptr(d_key_idx), _ptr(d_c), _ptr(m), _ptr(r), ctx.stream()), "mpe_paillier_open")
    return m, r


def lindell_pdl_proof(ctx, sk, stm, d_c_key, d_x1, d_r, nonces, d_key_idx=None, d_st_idx=None):
    """party one's `pdl_proof` (party_one.rs:366-401): returns (Q [B,16], proof dict)."""
    B = d_x1.shape[0]
    out = {f: _new(ctx, B, w) for f, w in PDL_PROOF_WORDS.items()}
    Q = _new(ctx, B, 16)
    nn, pr = _struct(N_.PdlNonces, nonces), _struct(N_.PdlProof, out)
    N_.check(N_.lib.mpe_lindell_pdl_proof(ctx.h, sk.h, stm.h, B, _ptr(d_key_idx), _ptr(d_st_idx), _ptr(d_c_key), _ptr(d_x1), _ptr(d_r),
                                          C.byref(nn), _ptr(Q), C.byref(pr), ctx.stream()), "mpe_lindell_pdl_proof")
    return Q, out


def lindell_pdl_verify(ctx, pk, d_Nt, d_h1, d_h2, d_dlog_x, d_dlog_y, d_stmt_N, d_stmt_c, d_stmt_Q, d_c_key, d_q1, proof, d_key_idx=None):
    """party two's `PaillierPublic::pdl_verify` (party_two.rs:275-300): ok flags [B]."""
    B = d_stmt_c.shape[0]
    ok = _flags(ctx, B)
    pr = _struct(N_.PdlProof, proof)
    N_.check(N_.lib.mpe_lindell_pdl_verify(ctx.h, pk.h, B, _ptr(d_key_idx), _ptr(d_Nt), _ptr(d_h1), _ptr(d_h2), _ptr(d_dlog_x), _ptr(d_dlog_y),
                                           _ptr(d_stmt_N), _ptr(d_stmt_c), _ptr(d_stmt_Q), _ptr(d_c_key), _ptr(d_q1), C.byref(pr), _ptr(ok),
                                           ctx.stream()), "mpe_lindell_pdl_verify")
    return ok


def lindell_sign(ctx, sk, d_c3, d_k1, d_R2, d_key_idx=None):
    """`Signature::compute_with_recid` batched: returns (r [B,8], s [B,8], recid [B]) on the device."""
    B = d_c3.shape[0]
    r, s = _new(ctx, B, 8), _new(ctx, B, 8)
    recid = torch.empty((B,), dtype=torch.int32, device=ctx.device)
    N_.check(N_.lib.mpe_lindell_sign(ctx.h, sk.h, B, _ptr(d_key_idx), _ptr(d_c3), _ptr(d_k1), _ptr(d_R2), _ptr(r), _ptr(s),
                                     _ptr(recid), ctx.stream()), "mpe_lindell_sign")
    return r, s, recid
