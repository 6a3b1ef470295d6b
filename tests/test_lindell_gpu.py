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


def test_lindell_keygen_pdl_exchange(gpu_ctx, keys):
    """party one's `pdl_proof` (party_one.rs:366-401) and party two's `PaillierPublic::pdl_verify` (party_two.rs:275-300):
    the proof bytes equal the oracle's PDLwSlackProof::prove on the same statement, the verdict equals
    (statement == what party two holds) && CompositeDLogProof::verify && PDLwSlackProof::verify of the oracle, under a
    tampered proof field, a wrong composite-dlog proof, and a statement that names another ciphertext / Q / key."""
    from multi_party_ecdsa_amd import engine as E
    import keygen_fixture as KG
    import orc
    B = 12
    r = F.Rng("gpu-lindell-pdl")
    nk = len(keys)
    kidx = [(5 * i + 1) % 4 for i in range(B)]                       # Paillier keys 0..3, statements from keys 4..
    sidx = [4 + (i % 6) for i in range(B)]
    x1 = [r.below(pyref.Q // 3) + 1 for _ in range(B)]
    r0 = [r.below(keys[k].N) for k in kidx]
    Ntab = F.words([k.N for k in keys], 64)
    c_key = orc.paillier_encrypt(Ntab, F.words(x1, 64), F.words(r0, 64), kidx)
    Qp = [pyref.ec_mul(x, pyref.G) for x in x1]
    Gw = F.point_words([pyref.G] * B)
    # the (N~, h1, h2) of every item and its CompositeDLogProof (minted by the oracle's prove side)
    cN, cg, cni, cx, cy = KG.composite_dlog_case([keys[s] for s in sidx], seed="lindell-cdlog")
    Nt, h1, h2 = F.ints(cN), F.ints(cg), F.ints(cni)

    class St:                                                        # statement i as a key-like object for the nonce sampler
        def __init__(self, n): self.Nt = n
    nn = [F.pdl_nonces(r, keys[k], St(n)) for k, n in zip(kidx, Nt)]
    nw = {f: F.words([n[f] for n in nn], w) for f, w in E.PDL_NONCE_WORDS.items()}
    dev = lambda a: _dev(gpu_ctx, a)
    di = lambda v: torch.tensor(v, dtype=torch.int32, device=gpu_ctx.device)
    sk = E.PaillierKeys(gpu_ctx, p=[k.p for k in keys], q=[k.q for k in keys])
    pk = E.PaillierKeys(gpu_ctx, N=[k.N for k in keys])
    stm = E.Statements(gpu_ctx, Nt, h1, h2)                           # party one's statement table (item i -> statement i)
    Q, pr = E.lindell_pdl_proof(gpu_ctx, sk, stm, dev(c_key), dev(F.words(x1, 8)), dev(F.words(r0, 64)), {f: dev(v) for f, v in nw.items()}, di(kidx))
    gpu_ctx.sync()
    npw = lambda t: t.cpu().numpy().view(np.uint32)
    assert np.array_equal(npw(Q), F.point_words(Qp))
    want = orc.pdl_prove(Ntab, cN, cg, cni, kidx, list(range(B)), c_key, F.point_words(Qp), Gw, F.words(x1, 8), F.words(r0, 64),
                         nw["alpha"], nw["beta"], nw["rho"], nw["gamma"])
    for f in want:
        assert np.array_equal(npw(pr[f]), want[f]), f

    def run(proof, dx=cx, dy=cy, sN=None, sc=c_key, sQ=None, q1=None):
        sN = Ntab[kidx] if sN is None else sN
        sQ = F.point_words(Qp) if sQ is None else sQ
        q1 = F.point_words(Qp) if q1 is None else q1
        ok = E.lindell_pdl_verify(gpu_ctx, pk, dev(cN), dev(cg), dev(cni), dev(dx), dev(dy), dev(sN), dev(sc), dev(sQ), dev(c_key), dev(q1),
                                  proof, di(kidx))
        gpu_ctx.sync()
        pw = {k: (npw(v) if hasattr(v, "cpu") else v) for k, v in proof.items()}
        w_pdl = orc.pdl_verify(Ntab, cN, cg, cni, kidx, list(range(B)), sc, sQ, Gw, pw)
        w_cd = np.zeros(B, dtype=np.uint8)
        orc.lib.orc_composite_dlog_verify(B, orc._p(cN), orc._p(cg), orc._p(cni), orc._p(np.ascontiguousarray(dx)), orc._p(np.ascontiguousarray(dy)), orc._p(w_cd))
        same = [int(np.array_equal(sN[i], Ntab[kidx[i]]) and np.array_equal(sc[i], c_key[i]) and np.array_equal(sQ[i], q1[i])) for i in range(B)]
        exp = [int(a and b and c) for a, b, c in zip(w_pdl, w_cd, same)]
        assert list(ok.cpu().numpy()) == exp
        return exp

    assert run(pr) == [1] * B
    bad = {k: v.clone() for k, v in pr.items()}
    bad["s2"][3, 0] ^= 1
    bad["u3"][5, 7] ^= 4
    assert run(bad) == [0 if i in (3, 5) else 1 for i in range(B)]
    y2 = cy.copy(); y2[1, 0] ^= 1
    x2 = cx.copy(); x2[6, 3] ^= 2
    assert run(pr, dx=x2, dy=y2) == [0 if i in (1, 6) else 1 for i in range(B)]
    sc2 = c_key.copy(); sc2[2, 5] ^= 1                                # the statement names another ciphertext
    q1b = F.point_words(Qp).copy(); q1b[4] = F.point_words([pyref.ec_mul(7, pyref.G)])[0]   # party two holds another Q1
    sN2 = Ntab[kidx].copy(); sN2[7] = Ntab[(kidx[7] + 1) % 4]        # ... another Paillier key
    assert run(pr, sN=sN2, sc=sc2, q1=q1b) == [0 if i in (2, 4, 7) else 1 for i in range(B)]
