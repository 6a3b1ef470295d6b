"""GPU: the reference's fault-injection matrix (gg_2020/test.rs:69-148) on the round engine (`mpe_gg20_session_fault_inject`
doubles delta_i / sigma_i / s_i of the chosen parties) and the batched blame entry points (`mpe_gg20_blame5/6/7` =
gg_2020/blame.rs): every round message stays byte-identical to the equally corrupted oracle, the same check fails with the
same status, and the blame functions name EXACTLY the corrupted set — on the GPU and on the oracle; ECDDHProof entry points."""
import numpy as np
import pytest
import torch

import fixtures as F
import gg20_fixture as G
import orc
import pyref
from test_blame_cpu import EXPECT_STATUS, MATRIX, openings6, openings7, run_oracle_with_faults

pytestmark = pytest.mark.gpu


def _dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("t,n,signers,step,corrupted", MATRIX)
def test_fault_injection_matrix(gpu_ctx, keys, t, n, signers, step, corrupted):
    from multi_party_ecdsa_amd import engine as E
    B, S = 2, len(signers)
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed=f"blame-gpu-{t}-{n}-{step}-{corrupted}")
    want_mask = sum(1 << i for i in corrupted)
    oparties, oslabs = run_oracle_with_faults(lk, nonces, B, step, corrupted)
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])
    dn = {f: _dev(gpu_ctx, v) for f, v in nonces.items()}
    sess = E.Gg20Session(gpu_ctx, gk, B, list(range(S)), dn)
    sess.fault_inject(step, want_mask)
    slabs, prev = {}, None
    for rnd in range(9):
        out = sess.round(rnd, d_in=prev, msg=dn["msg"] if rnd == 7 else None)
        if rnd in G.ROUNDS:
            slabs[rnd] = _u32(out)
            assert np.array_equal(slabs[rnd], oslabs[rnd]), f"round {rnd} differs from the equally corrupted oracle"
            prev = out
    res = sess.result()
    gpu_ctx.sync()
    assert (res["status"].cpu().numpy() == EXPECT_STATUS[step]).all()
    if step == 5:
        which, o = "b5", G.blame5_opened(lk, nonces, slabs, B)
        got = E.gg20_blame5(gpu_ctx, gk, B, {f: _dev(gpu_ctx, v) for f, v in o.items()})
    elif step == 6:
        en = F.words([F.Rng("ecddh-gpu").below(pyref.Q - 1) + 1 for _ in range(B * S)], 8)
        which, o = "b6", openings6(lk, nonces, oslabs, oparties, B, en)
        # what the GPU parties publish themselves: miu and the ECDDH proof (sigma_i never leaves the session object)
        miu, a1, a2, z = sess.blame6_state(_dev(gpu_ctx, en))
        tr = lambda x: np.ascontiguousarray(np.moveaxis(_u32(x), 0, 1)).reshape(-1, x.shape[-1])
        assert np.array_equal(tr(miu.reshape(S, B, -1)).reshape(-1, 64), o["miu"])
        assert np.array_equal(tr(a1), o["a1"]) and np.array_equal(tr(a2), o["a2"]) and np.array_equal(tr(z), o["z"])
        # ... and what each key holder opens of the w_i ciphertexts it received (Paillier::open, blame.rs:252-256)
        sg, P1 = [int(x) for x in lk["arrays"]["signers"]], S - 1
        sk_all = E.PaillierKeys(gpu_ctx, p=[k.p for k in lk["keys"]], q=[k.q for k in lk["keys"]])
        kx = torch.tensor([sg[(r // P1) % S] for r in range(B * S * P1)], dtype=torch.int32, device=gpu_ctx.device)
        om, orr = E.paillier_open(gpu_ctx, sk_all, _dev(gpu_ctx, o["c_b"]), kx)
        assert np.array_equal(_u32(om), o["miu"]) and np.array_equal(_u32(orr), o["miu_rand"])
        got = E.gg20_blame6(gpu_ctx, gk, B, {f: _dev(gpu_ctx, v) for f, v in o.items()})
    else:
        which, o = "b7", openings7(lk, nonces, slabs, oparties, B)
        got = E.gg20_blame7(gpu_ctx, S, B, {f: _dev(gpu_ctx, v) for f, v in o.items()})
    gpu_ctx.sync()
    assert list(_u32(got)) == [want_mask] * B == list(G.oracle_blame(lk, which, o, B))


def test_blame_on_lying_openings(gpu_ctx, keys):
    """Openings that do not match the public ciphertexts: the liar is named and the delta / sigma check is skipped, as in
    blame.rs:140,181 — session 0 honest openings, session 1 a wrong k, session 2 a wrong beta_tag, session 3 a wrong miu randomness"""
    from multi_party_ecdsa_amd import engine as E
    t, n, signers, B = 1, 3, [0, 2], 4
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed="blame-liars")
    oparties, oslabs = run_oracle_with_faults(lk, nonces, B, 5, [1])
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])
    o = G.blame5_opened(lk, nonces, oslabs, B)
    o["k"][1 * 2 + 0, 1] ^= 1                   # session 1: signer 0 lies about k
    o["beta_tag"][2 * 2 + 1, 5] ^= 8            # session 2: the beta_tag of the MessageB Alice 1 received (sent by signer 0)
    got = list(_u32(E.gg20_blame5(gpu_ctx, gk, B, {f: _dev(gpu_ctx, v) for f, v in o.items()})))
    assert got == list(G.oracle_blame(lk, "b5", o, B)) == [0b10, 0b01, 0b01, 0b10]
    oparties, oslabs = run_oracle_with_faults(lk, nonces, B, 6, [0])
    en = F.words([F.Rng("ecddh-liar").below(pyref.Q - 1) + 1 for _ in range(B * 2)], 8)
    o = openings6(lk, nonces, oslabs, oparties, B, en)
    o["miu_rand"][3 * 2 + 1, 0] ^= 1            # session 3: signer 1 opens a wrong randomness
    o["z"][1 * 2 + 1, 0] ^= 1                   # session 1: signer 1's ECDDH proof is broken too
    got = list(_u32(E.gg20_blame6(gpu_ctx, gk, B, {f: _dev(gpu_ctx, v) for f, v in o.items()})))
    assert got == list(G.oracle_blame(lk, "b6", o, B)) == [0b01, 0b11, 0b01, 0b10]


def test_ecddh_entry_points(gpu_ctx):
    from multi_party_ecdsa_amd import engine as E
    rg = F.Rng("ecddh-op")
    B = 6
    sc = lambda: F.words([rg.below(pyref.Q - 1) + 1 for _ in range(B)], 8)
    x, s = sc(), sc()
    g1, g2 = orc.ec_mul_base(sc()), orc.ec_mul_base(sc())
    h1, h2 = orc.ec_mul(x, g1), orc.ec_mul(x, g2)
    a1, a2, z = orc.u32((B, 16)), orc.u32((B, 16)), orc.u32((B, 8))
    orc.lib.orc_ecddh_prove(B, *[orc._p(a) for a in (x, s, g1, h1, g2, h2, a1, a2, z)])
    stt = {f: _dev(gpu_ctx, v) for f, v in dict(g1=g1, h1=h1, g2=g2, h2=h2).items()}
    pr = E.ecddh_prove(gpu_ctx, _dev(gpu_ctx, x), _dev(gpu_ctx, s), stt)
    gpu_ctx.sync()
    assert np.array_equal(_u32(pr["a1"]), a1) and np.array_equal(_u32(pr["a2"]), a2) and np.array_equal(_u32(pr["z"]), z)
    pr["z"][2, 3] ^= 1
    pr["a2"][4] = pr["a1"][4]
    ok = np.zeros(B, dtype=np.uint8)
    orc.lib.orc_ecddh_verify(B, *[orc._p(a) for a in (g1, h1, g2, h2, _u32(pr["a1"]), _u32(pr["a2"]), _u32(pr["z"]), ok)])
    assert list(E.ecddh_verify(gpu_ctx, stt, pr).cpu().numpy()) == list(ok) == [1, 1, 0, 1, 0, 1]


def test_paillier_open_matches_oracle(gpu_ctx, keys):
    """kzen-paillier `Open::open` (blame.rs:252-256): (m, r) of fresh encryptions come back exactly; edge ciphertexts
    (1, N + 1, N^2 - 1) give the oracle's bytes as well.  (Non-units modulo N^2 are outside Paillier's domain: the L
    function is not an exact division there and the reference's answer is an artefact of GMP's rounding.)"""
    import torch
    from multi_party_ecdsa_amd import engine as E
    import fixtures as F
    import orc
    r = F.Rng("gpu-open")
    B = 23
    kidx = [(7 * i) % len(keys) for i in range(B)]
    m = [r.below(keys[k].N) for k in kidx]
    rr = [r.coprime_below(keys[k].N) for k in kidx]
    m[0], m[1] = 0, keys[kidx[1]].N - 1
    c = [pyref.paillier_encrypt(keys[k].N, mm, x) for k, mm, x in zip(kidx, m, rr)]
    N5 = keys[kidx[5]].N
    hostile = {18: 1, 19: keys[kidx[19]].N + 1, 20: keys[kidx[20]].N ** 2 - 1, 21: 2, 22: keys[kidx[22]].N ** 2 - keys[kidx[22]].N - 1}
    for i, v in hostile.items():
        c[i] = v
    sk = E.PaillierKeys(gpu_ctx, p=[k.p for k in keys], q=[k.q for k in keys])
    di = torch.tensor(kidx, dtype=torch.int32, device=gpu_ctx.device)
    gm, gr = E.paillier_open(gpu_ctx, sk, E.dev(gpu_ctx, c, 128), di)
    gpu_ctx.sync()
    wm, wr = orc.u32((B, 64)), orc.u32((B, 64))
    orc.lib.orc_paillier_open(B, len(keys), orc._p(F.words([k.p for k in keys], 32)), orc._p(F.words([k.q for k in keys], 32)),
                              orc._p(np.array(kidx, dtype=np.int32)), orc._p(F.words(c, 128)), orc._p(wm), orc._p(wr))
    gmw, grw = gm.cpu().numpy().view(np.uint32), gr.cpu().numpy().view(np.uint32)
    assert [i for i in range(B) if not np.array_equal(gmw[i], wm[i])] == []
    assert [i for i in range(B) if not np.array_equal(grw[i], wr[i])] == []
    for i in range(18):
        assert F.ints(wm[i:i + 1])[0] == m[i] and F.ints(wr[i:i + 1])[0] == rr[i]


def test_blame_at_scale_the_bench_section(gpu_ctx, keys):
    """bench.py's `f2_blame_4096` section at 96 sessions: every session fails with the reference's status after the injected
    fault, the three blame calls name exactly the corrupted signer everywhere, the phase-6 openings a party derives on the device
    (Paillier::open of what it received) equal its own miu, and the first sessions equal the oracle's blame"""
    import bench
    from multi_party_ecdsa_amd import engine as E
    gen = torch.Generator(device=gpu_ctx.device)
    gen.manual_seed(7)
    res = bench.blame_section(gpu_ctx, E, G, keys, F, gen, B=96, cpu=True, sample=6)
    for ph, code in (("blame5", 502), ("blame6", 602), ("blame7", 701)):
        r = res[ph]
        assert r[f"failing_check_is_{code}_everywhere"] and r["names_exactly_the_corrupted_signer"] and r["parity_vs_oracle_on_sample"], (ph, r)
        assert r["sessions_per_s"] > 0
    assert res["blame6"]["paillier_open_equals_the_sessions_own_miu"]
