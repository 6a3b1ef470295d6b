"""Thin object layer over the C-ABI: device buffers are torch tensors (int32 storage of the
u32 words), every compute call goes to libmpecdsa_hip.so.  Mirrors curv's `BigInt::mod_pow` /
`mod_mul` in batched form."""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .words import ints_to_words, words_to_ints


def _dev_u32(arr_np, device):
    """np.uint32 [.., ..] -> torch int32 tensor on device holding the same bits."""
    return torch.from_numpy(arr_np.view(np.int32)).to(device)


def _to_np_u32(t):
    return t.detach().cpu().numpy().view(np.uint32)


class Context:
    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise N.MpeError("no GPU visible: the HIP path cannot run (there is no CPU fallback)")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        N.check(N.lib.mpe_ctx_create(C.byref(h), device), "mpe_ctx_create")
        self.h = h

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def sync(self):
        N.check(N.lib.mpe_sync(self.h, self.stream()), "mpe_sync")

    def launch_info(self):
        li = N.LaunchInfo()
        N.check(N.lib.mpe_last_launch_info(self.h, C.byref(li)), "mpe_last_launch_info")
        return {k: getattr(li, k) for k, _ in li._fields_}

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
