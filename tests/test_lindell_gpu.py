"""GPU parity tests of the Lindell'17 signing entry points (mpe_lindell_partial_sig / mpe_lindell_sign) against the
GMP oracle: byte-identical c3 and (r, s, recid) on seeded inputs with edge messages, ECDSA verification of the result
under the joint public key, and a ragged batch with per-item keys."""
import numpy as np
import pytest
import torch

import fixtures as F
import lindell_fixture as L
import pyref

pytestmark = pytest.mark.gpu


def _dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


@pytest.mark.parametrize("B", [1, 45])
def test_lindell_sign_matches_oracle(gpu_ctx, keys, B):
    from multi_party_ecdsa_amd import engine as E
    fx = L.make(keys, B, seed=f"gpu-lindell-{B}")
    want_c3, want_r, want_s, want_recid = L.oracle_run(fx)
    sk = E.PaillierKeys(gpu_ctx, p=[k.p for k in keys], q=[k.q for k in keys])       # party one
    pk = E.PaillierKeys(gpu_ctx, N=[k.N for k in keys])                              # what party two holds
    kidx = torch.tensor(fx["kidx"], dtype=torch.int32, device=gpu_ctx.device)
    d = lambda name: _dev(gpu_ctx, fx[name])
    c3 = E.lindell_partial_sig(gpu_ctx, pk, d("c_key"), d("x2"), d("k2"), d("R1"), d("msg"), d("rho"), d("r"), kidx)
    r, s, recid = E.lindell_sign(gpu_ctx, sk, c3, d("k1"), d("R2"), kidx)
    gpu_ctx.sync()
    assert np.array_equal(c3.cpu().numpy().view(np.uint32), want_c3)
    assert np.array_equal(r.cpu().numpy().view(np.uint32), want_r)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), want_s)
    assert list(recid.cpu().numpy()) == list(want_recid)
    for i in range(B):
        assert pyref.ecdsa_verify(fx["pub"][i], fx["msg_int"][i] % pyref.Q, F.ints(want_r[i:i + 1])[0], F.ints(want_s[i:i + 1])[0])


def test_lindell_bad_args(gpu_ctx, keys):
    from multi_party_ecdsa_amd import engine as E, _native as N
    pk = E.PaillierKeys(gpu_ctx, N=[k.N for k in keys])
    z = torch.zeros((1, 128), dtype=torch.int32, device=gpu_ctx.device)
    with pytest.raises(N.MpeError):                                                  # a public key set cannot decrypt
        E.lindell_sign(gpu_ctx, pk, z, z[:, :8].contiguous(), z[:, :16].contiguous())
