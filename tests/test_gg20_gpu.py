"""GPU parity tests of the GG20 signing engine against the CPU oracle (oracle/gg20_oracle.c):
 * the ROUND VIEW (`mpe_gg20_round0..7`, one `RoundN::proceed` per call): EVERY outgoing message slab byte-identical to
   the oracle's, with all parties local and with one party per object (which only ever holds that party's secrets);
 * tampered messages: the same status INTEGER and bad_actors as the oracle for every check of the protocol;
 * the lock-step composition `mpe_gg20_sign`: byte-identical (r, s, recid) and R for the reference's own
   (t, n, signer-set) cases (state_machine/sign.rs:740-763; gg_2020/test.rs:55-67 uses S > t+1 too); signatures
   re-checked by the independent Python ECDSA verifier (the reference's check_sig, gg_2020/test.rs:711-748)."""
import numpy as np
import pytest
import torch

import fixtures as F
import gg20_fixture as G
import pyref

pytestmark = pytest.mark.gpu


def _dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def _run(gpu_ctx, keys, t, n, signers, B, seed, **kw):
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed=seed)
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])
    out = E.gg20_sign(gpu_ctx, gk, {f: _dev(gpu_ctx, v) for f, v in nonces.items()}, B, want_R=True, **kw)
    gpu_ctx.sync()
    return lk, nonces, [o.cpu().numpy() for o in out]


class GpuParty:
    """adapter: a Gg20Session over the local parties `local` with numpy slabs in / out (test relay)"""

    def __init__(self, ctx, gk, B, local, nonces_np, **kw):
        from multi_party_ecdsa_amd import engine as E
        self.ctx = ctx
        self.keep = {f: _dev(ctx, v) for f, v in nonces_np.items()}
        self.sess = E.Gg20Session(ctx, gk, B, local, self.keep, **kw)

    def round(self, rnd, slab):
        if rnd == 7:
            out = self.sess.round(7, msg=_dev(self.ctx, slab))
        else:
            out = self.sess.round(rnd, d_in=None if slab is None else _dev(self.ctx, slab))
        self.ctx.sync()
        return None if out is None else _u32(out)

    def result(self):
        o = self.sess.result()
        self.ctx.sync()
        return {f: (_u32(v) if f in ("r", "s", "R", "bad_actors") else v.cpu().numpy()) for f, v in o.items()}


CASES = [(1, 3, [0, 1], 3), (2, 5, [0, 2, 4], 2), (1, 3, [0, 1, 2], 2),        # the third: S > t+1, as gg_2020/test.rs:55-67
         (3, 8, [1, 2, 5, 7], 1)]                                              # n = 8: the widest shape the library takes


@pytest.mark.parametrize("t,n,signers,B", CASES)
def test_every_round_message_is_byte_identical_all_parties_local(gpu_ctx, keys, t, n, signers, B):
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed=f"rounds-{t}-{n}-{signers}")
    want = G.oracle_sign_ex(lk, nonces, B)
    assert not want["status"].any()
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])
    S = len(signers)

    class All:
        def __init__(self):
            self.p = GpuParty(gpu_ctx, gk, B, list(range(S)), nonces)

        def round(self, rnd, slab):
            return self.p.round(rnd, slab)
    eng = All()
    prev = None
    for rnd in range(9):
        out = eng.round(rnd, nonces["msg"] if rnd == 7 else prev)
        if rnd in G.ROUNDS:
            assert out.shape == want["slabs"][rnd].shape
            assert np.array_equal(out, want["slabs"][rnd]), f"message slab of round {rnd}"
            prev = out
    res = eng.p.result()
    assert not res["status"].any() and not res["bad_actors"].any()
    for i in range(S):
        assert np.array_equal(res["r"][i], want["r"]) and np.array_equal(res["s"][i], want["s"]) and list(res["recid"][i]) == list(want["recid"])
        assert np.array_equal(res["R"][i], want["R"])


@pytest.mark.parametrize("t,n,signers,B", CASES[:2])
def test_one_party_per_object_holds_only_its_own_secrets(gpu_ctx, keys, t, n, signers, B):
    """The per-party view: S objects, each created from a key object that has ONE party's x_i, p_i, q_i, exchange the
    round messages through a relay; every message and every party's signature equal the oracle's."""
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed=f"party-{t}-{n}-{signers}")
    want = G.oracle_sign_ex(lk, nonces, B)
    S = len(signers)
    parties = []
    for i in range(S):
        gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"], own=[signers[i]])
        class One:
            def __init__(self, gk, i):
                self.gk, self.p = gk, GpuParty(gpu_ctx, gk, B, [i], G.party_nonces(nonces, lk, i))
            def round(self, rnd, slab):
                o = self.p.round(rnd, slab)
                return None if o is None else o[0]
            def result(self):
                return self.p.result()
        parties.append(One(gk, i))
    slabs = G.run_rounds(parties, nonces["msg"])
    for rnd in G.ROUNDS:
        assert np.array_equal(slabs[rnd], want["slabs"][rnd]), f"round {rnd}"
    for p in parties:
        res = p.result()
        assert not res["status"].any()
        assert np.array_equal(res["r"][0], want["r"]) and np.array_equal(res["s"][0], want["s"]) and list(res["recid"][0]) == list(want["recid"])


def test_parties_built_from_their_local_key_json(gpu_ctx, keys):
    """Each party's key object comes from its own LocalKey JSON (what the reference's gg20_keygen leaves on disk,
    keygen/rounds.rs:311-322) through multi_party_ecdsa_amd.wire: nobody sees another party's x_i, p, q; signatures = oracle's"""
    import json
    from multi_party_ecdsa_amd import engine as E, wire as W
    t, n, signers, B = 1, 3, [0, 2], 2
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed="local-key-json")
    want = G.oracle_sign_ex(lk, nonces, B)
    A = lk["arrays"]
    xs, X, y = F.ints(A["x"]), F.points(A["X"]), F.points(A["y"])[0]
    Ns, stm = [k.N for k in lk["keys"]], [(k.Nt, k.h1, k.h2) for k in lk["keys"]]
    parties = []
    for i, party in enumerate(signers):
        doc = json.dumps(W.local_key_to_json(party + 1, t, n, lk["keys"][party].p, lk["keys"][party].q, xs[party], y, X, Ns, stm))
        ka = W.local_keys_to_arrays([W.local_key_from_json(doc)])
        gk = E.Gg20Keys(gpu_ctx, ka["t"], ka["n"], signers, ka["arrays"], own=ka["own"])

        class One:
            def __init__(self, gk, i):
                self.gk, self.p = gk, GpuParty(gpu_ctx, gk, B, [i], G.party_nonces(nonces, lk, i))
            def round(self, rnd, slab):
                o = self.p.round(rnd, slab)
                return None if o is None else o[0]
        parties.append(One(gk, i))
    slabs = G.run_rounds(parties, nonces["msg"])
    for rnd in G.ROUNDS:
        assert np.array_equal(slabs[rnd], want["slabs"][rnd]), f"round {rnd}"
    for p_ in parties:
        res = p_.p.result()
        assert not res["status"].any() and np.array_equal(res["r"][0], want["r"]) and np.array_equal(res["s"][0], want["s"])


def _other_point(words16):
    """a different VALID point: the double of the given one"""
    P = F.points(words16.reshape(1, 16))[0]
    return F.point_words([pyref.ec_add(P, P)])[0]


# (round whose outgoing message is tampered, sender ordinal, word offset, kind) -> every check of the protocol
TAMPERS = [
    ("range proof s1", 0, 1, 136, "flip"),              # AliceProof st 0 of party 1 -> its peers fail MessageB::b: 101
    ("range proof z", 0, 0, 256 + 3, "flip"),           # statement 1
    ("MessageB b_proof.z", 1, 0, 160, "flip"),          # DLogProof::verify -> 201
    ("MessageB ciphertext", 1, 1, 5, "flip"),           # alpha wrong -> g^alpha check -> 201
    ("w MessageB beta_tag_proof.R", 1, 0, 208 + 184, "point"),
    ("TI != TIProof.com", 2, 1, 8, "point"),            # 303
    ("PedersenProof z1", 2, 0, 80, "flip"),             # 302
    ("PedersenProof a2", 2, 1, 48, "point"),            # 302
    ("decommit g_gamma", 3, 1, 8, "point"),             # 401, bad_actors = {1}
    ("decommit blind", 3, 0, 0, "flip"),                # 401, bad_actors = {0}
    ("PDL s2", 4, 0, 297, "flip"),                      # 501, bad_actors = {0}
    ("PDL u1", 4, 1, 64, "point"),                      # 501
    ("HEG z2", 5, 1, 56, "flip"),                       # 601, bad_actors = {1}
    ("S_i", 5, 0, 0, "point"),                          # 601 (and the sum)
    ("partial signature", 7, 1, 0, "flip"),             # 701
    # points that are not on the curve: curv would not deserialise the message -> 100*round + 90, bad_actors = the sender
    ("off-curve b_proof.pk", 1, 0, 128 + 2, "flip"),    # 290 at the receiver
    ("off-curve T_i", 2, 1, 8 + 9, "flip"),             # 390
    ("off-curve g_gamma", 3, 0, 8, "flip"),             # 490
    ("off-curve R_dash", 4, 1, 450 + 1, "flip"),        # 590 (S = 2: the last sub-record)
    ("off-curve PDL u1", 4, 0, 64 + 15, "flip"),        # 590
    ("off-curve S_i", 5, 1, 3, "flip"),                 # 690
    ("point at infinity as A3", 5, 0, 32, "zero"),      # 690
]


def test_tampered_messages_give_the_oracles_status_and_bad_actors(gpu_ctx, keys):
    """the whole matrix in ONE batch: session 0 clean, session k carries TAMPERS[k - 1] (one oracle run and one GPU run instead of 23 of each;
    the per-case form ran in the suite until round 6; tests/test_paths_gpu.py runs the same batch on the large-batch code paths)"""
    tamper_cases(gpu_ctx, keys, TAMPERS)


def tamper_cases(gpu_ctx, keys, cases):
    from multi_party_ecdsa_amd import engine as E
    t, n, signers, B = 1, 3, [0, 2], 1 + len(cases)
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed="tamper")
    S = len(signers)

    def tamper(r, slab):
        for k, (name, rnd, sender, word, kind) in enumerate(cases, start=1):      # session k only; session 0 stays clean
            if r != rnd:
                continue
            if kind == "zero":
                slab[sender, k, word:word + 16] = 0
            elif kind == "flip":
                slab[sender, k, word] ^= 4
            else:
                slab[sender, k, word:word + 16] = _other_point(slab[sender, k, word:word + 16])
    orc_parties = [G.OracleParty(lk, i, B, G.party_nonces(nonces, lk, i)) for i in range(S)]
    G.run_rounds(orc_parties, nonces["msg"], tamper)
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])

    class One:
        def __init__(self, i):
            self.p = GpuParty(gpu_ctx, gk, B, [i], G.party_nonces(nonces, lk, i))
        def round(self, r, slab):
            o = self.p.round(r, slab)
            return None if o is None else o[0]
    gpu_parties = [One(i) for i in range(S)]
    G.run_rounds(gpu_parties, nonces["msg"], tamper)
    seen = [set() for _ in range(B)]
    for i in range(S):
        w, g = orc_parties[i].result(), gpu_parties[i].p.result()
        for k in range(B):
            name = "clean" if k == 0 else cases[k - 1][0]
            assert int(g["status"][0][k]) == int(w["status"][k]), (name, i)
            assert int(g["bad_actors"][0][k]) == int(w["bad_actors"][k]), (name, i)
            seen[k].add(int(w["status"][k]))
        assert np.array_equal(g["r"][0], w["r"]) and np.array_equal(g["s"][0], w["s"])
        assert w["status"][0] == 0                                          # the clean session signs
    for k in range(1, B):
        assert seen[k] != {0}, (cases[k - 1][0], "the tampering must be detected by somebody")


@pytest.mark.parametrize("t,n,signers,B,kw", [
    (1, 2, [0, 1], 2, {}),                           # the reference's own three shapes, literally: gg_2020/test.rs:56-58,
    (4, 8, [0, 1, 2, 4, 6, 7], 1, {}),               # :65-67 (six of eight at t = 4) and (2, 5, [0, 2, 3, 4]) below (:61-63)
    (1, 3, [0, 1], 5, {}),
    (1, 3, [0, 2], 3, {"dedup_verify": True}),
    (1, 3, [1, 2], 7, {"chunk": 3}),                 # ragged chunking: 3 + 3 + 1
    (2, 5, [0, 2, 4], 3, {}),
    (2, 4, [1, 2, 3], 2, {"dedup_verify": True}),
    (2, 5, [0, 2, 3, 4], 2, {}),                     # S = 4 > t+1 (gg_2020/test.rs:60-63)
    # (n = 8 is the widest shape the library takes: (5, 8, six signers) ran in the suite until round 5 — 9.5 s — and all eight
    #  signing was run once: 18 s of oracle time, byte-identical; the reference's own (4, 8) shape above stays)
])
def test_sign_matches_oracle(gpu_ctx, keys, t, n, signers, B, kw):
    lk, nonces, (r, s, recid, status, R) = _run(gpu_ctx, keys, t, n, signers, B, f"gpu-{t}-{n}-{signers}", **kw)
    wr, ws, wrecid, wR, wstatus = G.oracle_sign(lk, nonces, B)
    assert list(wstatus) == [0] * B
    assert list(status) == [0] * B
    assert np.array_equal(r.view(np.uint32), wr) and np.array_equal(s.view(np.uint32), ws)
    assert list(recid) == list(wrecid)
    assert np.array_equal(R.view(np.uint32), wR)
    for b in range(B):
        m = F.ints(nonces["msg"][b:b + 1])[0]
        assert pyref.ecdsa_verify(lk["y"], m, F.ints(wr[b:b + 1])[0], F.ints(ws[b:b + 1])[0])


def test_large_batch_kernels_on_a_small_batch(keys):
    """Small batches take the latency-oriented variants (twice the lanes per exponentiation, lane groups in the EC
    round kernels); MPE_NO_ADAPTIVE_LANES forces the throughput variants the big batches use: same signatures."""
    import os
    from multi_party_ecdsa_amd import engine as E
    old = os.environ.get("MPE_NO_ADAPTIVE_LANES")
    os.environ["MPE_NO_ADAPTIVE_LANES"] = "1"
    try:
        ctx2 = E.Context(0)
    finally:
        if old is None:
            os.environ.pop("MPE_NO_ADAPTIVE_LANES", None)
        else:
            os.environ["MPE_NO_ADAPTIVE_LANES"] = old
    B = 4
    lk, nonces, (r, s, recid, status, R) = _run(ctx2, keys, 2, 5, [0, 1, 3], B, "gpu-throughput-variants")
    wr, ws, wrecid, wR, wstatus = G.oracle_sign(lk, nonces, B)
    assert list(status) == [0] * B == list(wstatus)
    assert np.array_equal(r.view(np.uint32), wr) and np.array_equal(s.view(np.uint32), ws) and list(recid) == list(wrecid)


def test_wrong_public_key_fails_only_that_check(gpu_ctx, keys):
    """phase6_check_S_i_sum: with an inconsistent y every session reports 602 on the GPU and on the oracle, no signature leaves"""
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    lk["arrays"]["y"][:] = F.point_words([pyref.ec_mul(999, pyref.G)])
    nonces = G.make_nonces(lk, 2, seed="gpu-bad-y")
    gk = E.Gg20Keys(gpu_ctx, 1, 3, [0, 1], lk["arrays"])
    r, s, recid, status = E.gg20_sign(gpu_ctx, gk, {f: _dev(gpu_ctx, v) for f, v in nonces.items()}, 2)
    gpu_ctx.sync()
    assert list(status.cpu().numpy()) == [602, 602] == list(G.oracle_sign(lk, nonces, 2)[4])
    assert not r.cpu().numpy().any() and not s.cpu().numpy().any()


def test_config4_sessions_byte_identical_to_the_threaded_oracle(gpu_ctx, keys):
    """BASELINE config 4's shape (t=1, n=3, signers {1,2}, one LocalKey, distinct nonces and messages per session) at
    a size the oracle finishes in seconds on the host cores: 192 sessions, every (r, s, recid, R) byte-identical."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    B = 192
    lk, nonces, (r, s, recid, status, R) = _run(gpu_ctx, keys, 1, 3, [0, 1], B, "gpu-config4")
    assert list(status) == [0] * B
    threads = max(1, min(os.cpu_count() or 1, 32))
    chunks = [c for c in np.array_split(np.arange(B), threads) if len(c)]
    outs = {}

    def run(ix):
        outs[int(ix[0])] = (len(ix), G.oracle_sign(lk, nonces, B, first=int(ix[0]), count=len(ix)))
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, chunks))
    for first, (cnt, (wr, ws, wrecid, wR, wstatus)) in outs.items():
        sl = slice(first, first + cnt)
        assert list(wstatus[sl]) == [0] * cnt
        assert np.array_equal(r.view(np.uint32)[sl], wr[sl]) and np.array_equal(s.view(np.uint32)[sl], ws[sl])
        assert list(recid[sl]) == list(wrecid[sl]) and np.array_equal(R.view(np.uint32)[sl], wR[sl])


def test_config5_share_t2n5_at_8192_sessions(gpu_ctx, keys):
    """BASELINE config 5's per-GPU share (t=2, n=5, three signers, 8 192 concurrent sessions): all signed; a sample checked
    against the oracle bit for bit (the same function bench.py reports as configs.c5_share_t2n5_8192)."""
    import bench
    from multi_party_ecdsa_amd import engine as E
    gen = torch.Generator(device=gpu_ctx.device)
    gen.manual_seed(55)
    res = bench.gg20_config(gpu_ctx, E, G, keys, 2, 5, 8192, 1, gen, parity_sample=24)
    assert res["all_sessions_signed"] and res["parity_vs_oracle_on_sample"]


def test_multi_wallet_batch_round_robin_16_key_sets(gpu_ctx, keys):
    """SURVEY.md 8d config 4 allows "16 fixtures round-robin": one batch whose sessions belong to different wallets
    (key sets): byte-identical to the oracle run with the same per-session key set."""
    from multi_party_ecdsa_amd import engine as E
    K, t, n, signers, B = 5, 1, 3, [0, 1], 11
    lks = [G.make_local_keys(keys[kk:] + keys[:kk], t, n, signers, seed=f"wallet-{kk}") for kk in range(K)]
    arrays = {f: np.concatenate([lk["arrays"][f] for lk in lks]) for f in ("x", "p", "q", "Nt", "h1", "h2", "y", "X")}
    arrays["signers"] = lks[0]["arrays"]["signers"]
    lkm = dict(lks[0], arrays=arrays, nkeysets=K)
    keyset = np.array([b % K for b in range(B)], dtype=np.int32)
    r0 = F.Rng("multi-wallet")
    # nonces must respect every wallet's own moduli: draw them per session from that wallet's fixture
    parts = [G.make_nonces(lks[int(keyset[b])], 1, seed=f"mw-{b}") for b in range(B)]
    nonces = {f: np.concatenate([p[f] for p in parts]) for f in parts[0]}
    want = G.oracle_sign_ex(lkm, nonces, B, keyset=keyset)
    assert not want["status"].any()
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, arrays, nkeysets=K)
    r, s, recid, status, R = E.gg20_sign(gpu_ctx, gk, {f: _dev(gpu_ctx, v) for f, v in nonces.items()}, B, want_R=True,
                                         keyset=torch.from_numpy(keyset).to(gpu_ctx.device))
    gpu_ctx.sync()
    assert list(status.cpu().numpy()) == [0] * B
    assert np.array_equal(_u32(r), want["r"]) and np.array_equal(_u32(s), want["s"]) and list(recid.cpu().numpy()) == list(want["recid"])
    for b in range(B):
        m = F.ints(nonces["msg"][b:b + 1])[0]
        assert pyref.ecdsa_verify(lks[int(keyset[b])]["y"], m, F.ints(want["r"][b:b + 1])[0], F.ints(want["s"][b:b + 1])[0])


def test_inconsistent_key_share_is_caught_with_the_oracles_status(gpu_ctx, keys):
    """A signer whose share x_i does not match its public X_i: the g^{w} check of rounds.rs:281 fails at its peer: 202 on
    the GPU and on the oracle, for every session."""
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, 1, 3, [0, 2])
    lk["arrays"]["x"][0, 0] ^= 1                                   # party 1's share off by a bit; X, y untouched
    B = 3
    nonces = G.make_nonces(lk, B, seed="gpu-bad-share")
    gk = E.Gg20Keys(gpu_ctx, 1, 3, [0, 2], lk["arrays"])
    r, s, recid, status = E.gg20_sign(gpu_ctx, gk, {f: _dev(gpu_ctx, v) for f, v in nonces.items()}, B)
    gpu_ctx.sync()
    want = G.oracle_sign(lk, nonces, B)[4]
    assert list(status.cpu().numpy()) == list(want) == [202] * B


def test_sigma_proofs_and_commitment_entry_points(gpu_ctx, keys):
    """mpe_pedersen_* / mpe_heg_* / mpe_hash_commit_point against the oracle's restatement of the curv proofs"""
    import orc
    from multi_party_ecdsa_amd import engine as E
    rg = F.Rng("sigma")
    B = 9
    sc = lambda: F.words([rg.below(pyref.Q - 1) + 1 for _ in range(B)], 8)
    m, r, s1, s2 = sc(), sc(), sc(), sc()
    m[0] = 0                                                        # x = 0: the HEG special case z1 = s1
    o = {f: orc.u32((B, w)) for f, w in dict(com=16, e=8, a1=16, a2=16, z1=8, z2=8).items()}
    orc.lib.orc_pedersen_prove(B, *[orc._p(a) for a in (m, r, s1, s2, o["com"], o["e"], o["a1"], o["a2"], o["z1"], o["z2"])])
    g = E.pedersen_prove(gpu_ctx, *[_dev(gpu_ctx, a) for a in (m, r, s1, s2)])
    gpu_ctx.sync()
    for f in o:
        assert np.array_equal(_u32(g[f]), o[f]), f
    g["z1"][3, 0] ^= 1
    assert list(E.pedersen_verify(gpu_ctx, g).cpu().numpy()) == [1, 1, 1, 0, 1, 1, 1, 1, 1]
    pts = lambda: orc.ec_mul_base(sc())
    Gp, H, Y = pts(), pts(), pts()
    D = orc.ec_add(orc.ec_mul(m, H), orc.ec_mul(r, Y))               # D = x H + r Y, E = r G
    Ee = orc.ec_mul(r, Gp)
    ho = {f: orc.u32((B, w)) for f, w in dict(T=16, A3=16, z1=8, z2=8).items()}
    orc.lib.orc_heg_prove(B, *[orc._p(a) for a in (m, r, s1, s2, Gp, H, Y, D, Ee, ho["T"], ho["A3"], ho["z1"], ho["z2"])])
    stt = {f: _dev(gpu_ctx, a) for f, a in dict(G=Gp, H=H, Y=Y, D=D, E=Ee).items()}
    hg = E.heg_prove(gpu_ctx, *[_dev(gpu_ctx, a) for a in (m, r, s1, s2)], stt)
    gpu_ctx.sync()
    for f in ho:
        assert np.array_equal(_u32(hg[f]), ho[f]), f
    ok = np.zeros(B, dtype=np.uint8)
    orc.lib.orc_heg_verify(B, *[orc._p(a) for a in (Gp, H, Y, D, Ee, ho["T"], ho["A3"], ho["z1"], ho["z2"], ok)])
    assert ok.all()
    hg["z2"][5, 2] ^= 8
    assert list(E.heg_verify(gpu_ctx, stt, hg).cpu().numpy()) == [1, 1, 1, 1, 1, 0, 1, 1, 1]
    blind = F.words([rg.bits(256) for _ in range(B)], 8)
    blind[1] = 0; blind[2, 7] = 0; blind[2, 6] &= 0xFF               # short blinding factors: minimal-length bytes
    com = orc.u32((B, 8))
    orc.lib.orc_hash_commit_point(B, orc._p(Gp), orc._p(blind), orc._p(com))
    assert np.array_equal(_u32(E.hash_commit_point(gpu_ctx, _dev(gpu_ctx, Gp), _dev(gpu_ctx, blind))), com)


def test_scratch_is_wiped_after_signing(keys):
    """Secret hygiene (the reference zeroizes its round-1 secrets, range_proofs.rs:26-36): after mpe_gg20_sign the session
    arena (k_i, gamma_i, w_i, sigma_i, ...), the message slabs, the composite workspace and the window tables of this
    context read back as zeros — and they did hold data while a session object was alive."""
    from multi_party_ecdsa_amd import engine as E
    ctx = E.Context(0)
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    nonces = G.make_nonces(lk, 2, seed="wipe")
    gk = E.Gg20Keys(ctx, 1, 3, [0, 1], lk["arrays"])
    dn = {f: _dev(ctx, v) for f, v in nonces.items()}
    sess = E.Gg20Session(ctx, gk, 2, [0, 1], dn)
    sess.round(0)
    nz, tot = ctx.scratch_audit()
    assert nz > 0 and tot > 0
    sess.close()
    r, s, recid, status = E.gg20_sign(ctx, gk, dn, 2)
    ctx.sync()
    assert list(status.cpu().numpy()) == [0, 0]
    nz, tot = ctx.scratch_audit()
    assert nz == 0 and tot > 0


def test_sessions_at_the_edges_of_the_sampling_ranges(gpu_ctx, keys):
    """the edge sessions of tests/test_oracle_edges_cpu.py (blinding factor 0, scalars 1 and q - 1, Paillier randomness 1, messages 0 /
    q / 2^256 - 1, zero proof nonces, a beta_tag that makes the MtA wrap) as ONE batch on the round engine: every round message of
    every party byte-identical to the oracle's, same statuses and bad actors, same signatures"""
    from multi_party_ecdsa_amd import engine as E
    import test_oracle_edges_cpu as TE
    t, n, signers = 1, 3, [0, 2]
    lk = G.make_local_keys(keys, t, n, signers)
    names = list(TE.EDGE_SESSIONS)
    parts = [TE._session_with(lk, "edge-" + nm, TE.EDGE_SESSIONS[nm]) for nm in names]
    nonces = {f: np.concatenate([p[f] for p in parts]) for f in parts[0]}
    B, S = len(names), len(signers)
    want = G.oracle_sign_ex(lk, nonces, B)
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])
    dn = {f: _dev(gpu_ctx, v) for f, v in nonces.items()}
    sess = E.Gg20Session(gpu_ctx, gk, B, list(range(S)), dn)
    prev = None
    for rnd in range(9):
        out = sess.round(rnd, d_in=prev, msg=dn["msg"] if rnd == 7 else None)
        if out is not None:
            assert np.array_equal(_u32(out), want["slabs"][rnd]), f"round {rnd}"
            prev = out.reshape(-1)
    res = sess.result()
    gpu_ctx.sync()
    st = res["status"].cpu().numpy()
    assert np.array_equal(st, want["party_status"]) and np.array_equal(_u32(res["bad_actors"]), want["party_bad"])
    failed = [names[b] for b in range(B) if st[:, b].any()]
    assert failed == list(TE.EXPECTED_FAILURES)
    for i in range(S):
        assert np.array_equal(_u32(res["r"])[i], want["r"]) and np.array_equal(_u32(res["s"])[i], want["s"])
