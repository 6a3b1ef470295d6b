"""GPU parity test of the MtA entry points (src/utilities/mta/mod.rs:52-179) against a composition of the
oracle's primitives, plus the reference's own check alpha + beta == a*b (mta/test.rs:6-19) and the
InvalidKey path (a tampered range proof)."""
import numpy as np
import pytest
import torch

import fixtures as F
import orc
import pyref

pytestmark = pytest.mark.gpu


def npw(t):
    return np.ascontiguousarray(t.cpu().numpy().view(np.uint32))


def test_mta_roundtrip_vs_oracle(gpu_ctx, keys):
    from multi_party_ecdsa_amd import engine as E
    r = F.Rng("gpu-mta-api")
    B, nst = 7, 3
    alice_keys = keys[:2]
    sk = E.PaillierKeys(gpu_ctx, p=[k.p for k in alice_keys], q=[k.q for k in alice_keys])
    stm = E.Statements(gpu_ctx, [k.Nt for k in keys[4:4 + nst]], [k.h1 for k in keys[4:4 + nst]], [k.h2 for k in keys[4:4 + nst]])
    kidx = [i % 2 for i in range(B)]
    di = lambda v: torch.tensor(v, dtype=torch.int32, device=gpu_ctx.device)
    a = [r.below(pyref.Q) for _ in range(B)]
    b = [r.below(pyref.Q) for _ in range(B)]
    ra = [r.below(alice_keys[k].N) for k in kidx]
    nn = [F.alice_nonces(r, alice_keys[kidx[i]], keys[4 + s]) for i in range(B) for s in range(nst)]
    nw = {f: F.words([n[f] for n in nn], w) for f, w in E.ALICE_NONCE_WORDS.items()}
    to_dev = lambda arr: torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(gpu_ctx.device)
    # ---- MessageA ----
    c_a, proofs = E.mta_message_a(gpu_ctx, sk, stm, E.dev(gpu_ctx, a, 8), E.dev(gpu_ctx, ra, 64), {f: to_dev(v) for f, v in nw.items()}, di(kidx))
    N = F.words([k.N for k in alice_keys], 64)
    tabs = [F.words([getattr(k, f) for k in keys[4:4 + nst]], 64) for f in ("Nt", "h1", "h2")]
    w_ca = orc.paillier_encrypt(N, F.words(a, 64), F.words(ra, 64), kidx)
    assert np.array_equal(npw(c_a), w_ca)
    kit, sit, bit = [kidx[i] for i in range(B) for _ in range(nst)], [s for _ in range(B) for s in range(nst)], [i for i in range(B) for _ in range(nst)]
    w_pr = orc.alice_generate(N, *tabs, kit, sit, F.words([a[i] for i in bit], 8), w_ca[bit], F.words([ra[i] for i in bit], 64),
                              nw["alpha"], nw["beta"], nw["gamma"], nw["rho"])
    for f in w_pr:
        assert np.array_equal(npw(proofs[f]), w_pr[f]), f
    # ---- MessageB ----
    bt = [r.below(alice_keys[k].N) for k in kidx]
    rb = [r.below(alice_keys[k].N) for k in kidx]
    nb, nbt = [r.below(pyref.Q - 1) + 1 for _ in range(B)], [r.below(pyref.Q - 1) + 1 for _ in range(B)]
    bad = {f: v.clone() for f, v in proofs.items()}
    bad["s"][2 * nst + 1, 3] ^= 1                           # exchange 2: one of its range proofs is invalid
    mb = E.mta_message_b(gpu_ctx, sk, stm, E.dev(gpu_ctx, b, 8), c_a, bad, E.dev(gpu_ctx, rb, 64), E.dev(gpu_ctx, bt, 64),
                         E.dev(gpu_ctx, nb, 8), E.dev(gpu_ctx, nbt, 8), di(kidx))
    exp_ok = [1] * B
    exp_ok[2] = 0                                           # Err(InvalidKey)  (mta/mod.rs:123-131)
    assert list(mb["ok"].cpu().numpy()) == exp_ok
    w_cbt = orc.paillier_encrypt(N, F.words(bt, 64), F.words(rb, 64), kidx)
    w_cb = orc.paillier_add(N, orc.paillier_mul(N, w_ca, F.words(b, 64), kidx), w_cbt, kidx)
    assert np.array_equal(npw(mb["c"]), w_cb)
    assert E.host(mb["beta"]) == [(-x) % pyref.Q for x in bt]
    wp = orc.dlog_prove(F.words(b, 8), F.words(nb, 8))
    wt = orc.dlog_prove(F.words([x % pyref.Q for x in bt], 8), F.words(nbt, 8))
    for got, want in ((mb["b_proof"], wp), (mb["beta_tag_proof"], wt)):
        assert np.array_equal(npw(got["pk"]), want[0]) and np.array_equal(npw(got["R"]), want[1]) and np.array_equal(npw(got["z"]), want[2])
    # ---- verify_proofs_get_alpha ----
    tam = {k: v.clone() for k, v in mb["beta_tag_proof"].items()}
    tam["z"][4, 0] ^= 1                                     # exchange 4: broken DLogProof
    alpha, share, ok = E.mta_verify_get_alpha(gpu_ctx, sk, mb["c"], mb["b_proof"], tam, E.dev(gpu_ctx, a, 8), di(kidx))
    exp = [1] * B
    exp[4] = 0
    assert list(ok.cpu().numpy()) == exp
    w_share = orc.paillier_decrypt(F.words([k.p for k in alice_keys], 32), F.words([k.q for k in alice_keys], 32), w_cb, kidx)
    assert np.array_equal(npw(share), w_share)
    al = E.host(alpha)
    assert al == [x % pyref.Q for x in F.ints(w_share)]
    # the reference's test: alpha + beta == a*b (mod q)   (mta/test.rs:16-18)
    assert [(x + y) % pyref.Q for x, y in zip(al, E.host(mb["beta"]))] == [x * y % pyref.Q for x, y in zip(a, b)]
