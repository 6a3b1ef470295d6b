"""GPU: mpe_correct_key_verify / mpe_composite_dlog_verify / mpe_vss_* against the oracle (gg_2020/party_i.rs:260-438)."""
import numpy as np
import pytest
import torch

import fixtures as F
import keygen_fixture as KF
import orc

pytestmark = pytest.mark.gpu


def _dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


def test_correct_key_verify_vs_oracle(gpu_ctx, keys):
    from multi_party_ecdsa_amd import engine as E
    N, sigma = KF.correct_key_case(keys)                                 # 16 keys x 11 roots
    sigma[11 * 3 + 7, 5] ^= 2
    sigma[11 * 9, 0] ^= 1
    N[12] = F.words([keys[12].p * 6361], 64)[0]                          # small prime factor
    want = np.zeros(16, dtype=np.uint8)
    orc.lib.orc_correct_key_verify(16, orc._p(N), orc._p(sigma), orc._p(want))
    got = E.correct_key_verify(gpu_ctx, _dev(gpu_ctx, N), _dev(gpu_ctx, sigma)).cpu().numpy()
    assert list(got) == list(want) and list(want) == [0 if i in (3, 9, 12) else 1 for i in range(16)]


def test_composite_dlog_verify_vs_oracle(gpu_ctx, keys):
    from multi_party_ecdsa_amd import engine as E
    N, g, ni, x, y = KF.composite_dlog_case(keys)
    y[2, 70] ^= 1
    x[5, 0] ^= 1
    ni[7] = F.words([keys[7].Nt - 1], 64)[0]                             # a different statement: proof no longer matches
    g[11] = F.words([0], 64)[0]                                          # gcd(g, N) != 1
    want = np.zeros(16, dtype=np.uint8)
    orc.lib.orc_composite_dlog_verify(16, *[orc._p(a) for a in (N, g, ni, x, y, want)])
    got = E.composite_dlog_verify(gpu_ctx, *[_dev(gpu_ctx, a) for a in (N, g, ni, x, y)]).cpu().numpy()
    assert list(got) == list(want) and list(want) == [0 if i in (2, 5, 7, 11) else 1 for i in range(16)]


def test_feldman_vs_oracle(gpu_ctx):
    from multi_party_ecdsa_amd import engine as E
    t, n, B = 2, 5, 9
    commits, shares, index, _ = KF.vss_case(t, n, B, seed="vss-gpu")
    shares[7, 1] ^= 1
    commits[20, 3] ^= 1                                                  # an off-curve commitment
    want = np.zeros(B * n, dtype=np.uint8)
    orc.lib.orc_vss_validate_share(B * n, t + 1, orc._p(commits), orc._p(shares), orc._p(index), orc._p(want))
    got = E.vss_validate_share(gpu_ctx, t + 1, _dev(gpu_ctx, commits), _dev(gpu_ctx, shares), _dev(gpu_ctx, index)).cpu().numpy()
    assert list(got) == list(want)
    commits[20, 3] ^= 1
    out = orc.u32((B * n, 16))
    orc.lib.orc_vss_point_commitment(B * n, t + 1, orc._p(commits), orc._p(index), orc._p(out))
    gp = E.vss_point_commitment(gpu_ctx, t + 1, _dev(gpu_ctx, commits), _dev(gpu_ctx, index)).cpu().numpy().view(np.uint32)
    assert np.array_equal(gp, out)


def test_keygen_prove_side_vs_oracle(gpu_ctx, keys):
    """NiCorrectKeyProof::proof and CompositeDLogProof::prove (party_i.rs:219-258): the bytes the oracle's prove side gives,
    and they verify on the GPU"""
    from multi_party_ecdsa_amd import engine as E
    import pyref
    N, want_sigma = KF.correct_key_case(keys)
    sk = E.PaillierKeys(gpu_ctx, p=[k.p for k in keys], q=[k.q for k in keys])
    sigma = E.correct_key_prove(gpu_ctx, sk)
    gpu_ctx.sync()
    got = sigma.cpu().numpy().view(np.uint32).reshape(-1, 64)
    assert np.array_equal(got, want_sigma)
    assert list(E.correct_key_verify(gpu_ctx, _dev(gpu_ctx, N), sigma.reshape(-1, 64)).cpu().numpy()) == [1] * len(keys)
    # composite dlog: same statements, secrets and nonces as the fixture -> same (x, y)
    r = F.Rng("cdlog")
    Ns, g, sec, nonce = [], [], [], []
    for k in keys:
        Ns.append(k.Nt); g.append(k.h1); sec.append(r.below(k.Nt >> 2)); nonce.append(r.bits(512))
    Nw, gw, nw, wx, wy = KF.composite_dlog_case(keys)
    x, y = E.composite_dlog_prove(gpu_ctx, _dev(gpu_ctx, Nw), _dev(gpu_ctx, gw), _dev(gpu_ctx, nw), _dev(gpu_ctx, F.words(sec, 64)),
                                  _dev(gpu_ctx, F.words(nonce, 16)))
    gpu_ctx.sync()
    assert np.array_equal(x.cpu().numpy().view(np.uint32), wx) and np.array_equal(y.cpu().numpy().view(np.uint32), wy)
    assert list(E.composite_dlog_verify(gpu_ctx, _dev(gpu_ctx, Nw), _dev(gpu_ctx, gw), _dev(gpu_ctx, nw), x, y).cpu().numpy()) == [1] * len(keys)


def test_keygen_verification_at_scale_the_bench_section(gpu_ctx, keys):
    """bench.py's `f3_keygen_verify_8192` section at 800 items: exactly the corrupted 1 % are refused by each of the three checks"""
    import bench
    from multi_party_ecdsa_amd import engine as E
    res = bench.keygen_verify_section(gpu_ctx, E, keys, F, B=800, cpu=True)
    assert res["corrupted"] == 8
    for f in ("correct_key", "composite_dlog", "vss"):
        assert res[f + "_exactly_the_corrupted_refused"], (f, res)
    assert res["oracle_accepts_the_uncorrupted_items"]


def test_round1_verdict_as_the_reference_composes_it_gpu(gpu_ctx, keys):
    """mpe_keygen_verify_round1 == oracle/gg20_oracle.c:orc_keygen_verify_round1 == party_i.rs:260-320: commitment, NiCorrectKeyProof,
    PAILLIER_MIN/MAX_BIT_LENGTH on e.n and dlog_statement.N, both CompositeDLogProofs, bad_actors — and the reference's
    test_small_paillier (gg_2020/test.rs:764-783): a 2046-bit key with a valid proof is refused"""
    from multi_party_ecdsa_amd import engine as E
    n = 3
    c = KF.round1_case(keys, n, 4)
    _, _, Nsmall, sig_small = KF.small_paillier_key()
    t = {f: a.copy() for f, a in c.items()}
    t["com"][0, 3] ^= 1
    t["blind"][4, 0] ^= 1
    t["N"][5], t["sigma"][5] = Nsmall[0], sig_small[0]
    t["sigma"][6, 70] ^= 1
    t["y_h1"][7, 2] ^= 1
    t["x_h2"][8, 9] ^= 1
    t["Nt"][9, 63] &= 0x3fffffff
    for case in (c, t):
        want_ok, want_bad = KF.oracle_round1(case, n)
        ok, bad = E.keygen_verify_round1(gpu_ctx, n, {f: _dev(gpu_ctx, a) for f, a in case.items()})
        assert list(ok.cpu().numpy()) == list(want_ok) and list(bad.cpu().numpy().view(np.uint32)) == list(want_bad)
    assert list(want_ok) == [0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1] and list(want_bad) == [0b001, 0b110, 0b111, 0b001]
    one = {f: a[5:6].copy() for f, a in t.items()}                      # test_small_paillier: share_count = 1, the party checks itself
    ok, bad = E.keygen_verify_round1(gpu_ctx, 1, {f: _dev(gpu_ctx, a) for f, a in one.items()})
    assert list(ok.cpu().numpy()) == [0] and list(bad.cpu().numpy()) == [1]
    # the proof of the small key is valid by itself: only the composed verdict stands in its way
    assert list(E.correct_key_verify(gpu_ctx, _dev(gpu_ctx, Nsmall), _dev(gpu_ctx, sig_small)).cpu().numpy()) == [1]


def test_round2_verdict_as_the_reference_composes_it_gpu(gpu_ctx):
    from multi_party_ecdsa_amd import engine as E
    t, n, B = 1, 3, 4
    commits, shares, index, _ = KF.vss_case(t, n, B, seed="vss-r2")
    y = np.ascontiguousarray(commits[:, :16]).copy()
    shares[2, 0] ^= 1
    y[7, 1] ^= 1
    want_ok, want_bad = np.zeros(B * n, dtype=np.uint8), np.zeros(B, dtype=np.uint32)
    orc.lib.orc_keygen_verify_round2(B * n, n, t + 1, orc._p(commits), orc._p(shares), orc._p(index), orc._p(y), orc._p(want_ok), orc._p(want_bad))
    ok, bad = E.keygen_verify_round2(gpu_ctx, n, t + 1, _dev(gpu_ctx, commits), _dev(gpu_ctx, shares), torch.from_numpy(index).to(gpu_ctx.device), _dev(gpu_ctx, y))
    assert list(ok.cpu().numpy()) == list(want_ok) and list(bad.cpu().numpy().view(np.uint32)) == list(want_bad) == [0b100, 0, 0b010, 0]
