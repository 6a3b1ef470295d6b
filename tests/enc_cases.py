"""One small bundle of inputs for every transcript whose byte-level conventions are recalled (curv's DLogProof, PedersenProof,
HomoELGamalProof, ECDDHProof, the HashCommitment; zk-paillier's NiCorrectKeyProof and CompositeDLogProof), with the outputs
the pure-Python restatement (tests/pyref*.py) gives under the encoding profile IN FORCE (pyref.ENC) and the same computation
on the C oracle.  tests/test_encodings_cpu.py compares the two under every profile of enc_profiles.PROFILES; the GPU test
(tests/test_encodings_gpu.py) compares the HIP engine with both.  Test infrastructure."""
import numpy as np

import fixtures as F
import orc
import pyref as R
import pyref_gg20 as PG


def inputs(keys, seed="enc-cases"):
    r = F.Rng(seed)
    rs = lambda: r.below(R.Q - 1) + 1
    k, st = keys[2], keys[9]
    a, l = rs(), rs()
    Rp = R.ec_mul(rs(), R.G)
    T = R.ec_add(R.ec_mul(a, R.G), R.ec_mul(l, R.H2))
    secret = r.below(st.Nt >> 2)
    ni = pow(pow(st.h1, secret, st.Nt), -1, st.Nt)
    return dict(
        dlog=dict(sk=rs(), nonce=rs()),
        pedersen=dict(m=a, r=l, s1=rs(), s2=rs()),
        heg=dict(x=l, r=a, s1=rs(), s2=rs(), G=Rp, D=T, E=R.ec_mul(a, Rp)),
        ecddh=dict(x=a, s=rs(), h1=R.ec_mul(a, R.G), g2=Rp, h2=R.ec_mul(a, Rp)),
        commit=[dict(point=R.ec_mul(rs(), R.G), blind=0), dict(point=R.ec_mul(rs(), R.G), blind=r.bits(256))],      # blind = 0: BigInt::to_bytes(0)
        ck=dict(p=k.p, q=k.q, N=k.N),
        cdlog=dict(N=st.Nt, g=st.h1, ni=ni, secret=secret, r=r.bits(512)),
    )


def python_outputs(inp):
    """under pyref.ENC"""
    d, p, h, e, c = inp["dlog"], inp["pedersen"], inp["heg"], inp["ecddh"], inp["cdlog"]
    return dict(
        dlog=R.dlog_prove(d["sk"], d["nonce"]),
        pedersen=PG.pedersen_prove(p["m"], p["r"], p["s1"], p["s2"]),
        heg=PG.heg_prove(h["x"], h["r"], h["s1"], h["s2"], h["G"], R.H2, R.G, h["D"], h["E"]),
        ecddh=PG.ecddh_prove(e["x"], e["s"], R.G, e["h1"], e["g2"], e["h2"]),
        commit=[PG.hash_commitment(x["point"], x["blind"]) for x in inp["commit"]],
        ck_sigma=R.correct_key_prove(inp["ck"]["p"], inp["ck"]["q"]),
        cdlog=R.composite_dlog_prove(c["N"], c["g"], c["ni"], c["secret"], c["r"]),
    )


W1 = lambda v, n: F.words([v], n)
P1 = lambda pt: F.point_words([pt])


def oracle_outputs(inp):
    """the same on the C oracle, under the profile last installed with orc.set_encoding (enc_profiles.applied sets both)"""
    d, p, h, e, c = inp["dlog"], inp["pedersen"], inp["heg"], inp["ecddh"], inp["cdlog"]
    _p = orc._p
    pk, Rr, z = orc.dlog_prove(W1(d["sk"], 8), W1(d["nonce"], 8))
    com, ee, a1, a2, z1, z2 = orc.u32((1, 16)), orc.u32((1, 8)), orc.u32((1, 16)), orc.u32((1, 16)), orc.u32((1, 8)), orc.u32((1, 8))
    orc.lib.orc_pedersen_prove(1, *[_p(a) for a in (W1(p["m"], 8), W1(p["r"], 8), W1(p["s1"], 8), W1(p["s2"], 8), com, ee, a1, a2, z1, z2)])
    ped = dict(e=F.ints(ee)[0], a1=F.points(a1)[0], a2=F.points(a2)[0], com=F.points(com)[0], z1=F.ints(z1)[0], z2=F.ints(z2)[0])
    T, A3, hz1, hz2 = orc.u32((1, 16)), orc.u32((1, 16)), orc.u32((1, 8)), orc.u32((1, 8))
    orc.lib.orc_heg_prove(1, *[_p(a) for a in (W1(h["x"], 8), W1(h["r"], 8), W1(h["s1"], 8), W1(h["s2"], 8), P1(h["G"]), P1(R.H2), P1(R.G),
                                               P1(h["D"]), P1(h["E"]), T, A3, hz1, hz2)])
    heg = dict(T=F.points(T)[0], A3=F.points(A3)[0], z1=F.ints(hz1)[0], z2=F.ints(hz2)[0])
    ea1, ea2, ez = orc.u32((1, 16)), orc.u32((1, 16)), orc.u32((1, 8))
    orc.lib.orc_ecddh_prove(1, *[_p(a) for a in (W1(e["x"], 8), W1(e["s"], 8), P1(R.G), P1(e["h1"]), P1(e["g2"]), P1(e["h2"]), ea1, ea2, ez)])
    coms = []
    for x in inp["commit"]:
        cw = orc.u32((1, 8))
        orc.lib.orc_hash_commit_point(1, _p(P1(x["point"])), _p(W1(x["blind"], 8)), _p(cw))
        coms.append(F.ints(cw)[0])
    sigma = orc.u32((11, 64))
    orc.lib.orc_correct_key_prove(1, _p(W1(inp["ck"]["p"], 32)), _p(W1(inp["ck"]["q"], 32)), _p(sigma))
    cx, cy = orc.u32((1, 64)), orc.u32((1, 73))
    orc.lib.orc_composite_dlog_prove(1, *[_p(a) for a in (W1(c["N"], 64), W1(c["g"], 64), W1(c["ni"], 64), W1(c["secret"], 64), W1(c["r"], 16), cx, cy)])
    return dict(dlog=(F.points(pk)[0], F.points(Rr)[0], F.ints(z)[0]), pedersen=ped, heg=heg,
                ecddh=(F.points(ea1)[0], F.points(ea2)[0], F.ints(ez)[0]), commit=coms, ck_sigma=F.ints(sigma),
                cdlog=(F.ints(cx)[0], F.ints(cy)[0]))


def oracle_verdicts(inp, out):
    """does the C oracle (under ITS current profile) accept the proofs in `out`?  -> dict of bools"""
    _p = orc._p
    h, e, c = inp["heg"], inp["ecddh"], inp["cdlog"]
    ok = np.zeros(1, dtype=np.uint8)
    res = {}
    pk, Rr, z = out["dlog"]
    res["dlog"] = bool(orc.dlog_verify(P1(pk), P1(Rr), W1(z, 8))[0])
    pe = out["pedersen"]
    orc.lib.orc_pedersen_verify(1, *[_p(a) for a in (P1(pe["com"]), P1(pe["a1"]), P1(pe["a2"]), W1(pe["z1"], 8), W1(pe["z2"], 8), ok)])
    res["pedersen"] = bool(ok[0])
    hg = out["heg"]
    orc.lib.orc_heg_verify(1, *[_p(a) for a in (P1(h["G"]), P1(R.H2), P1(R.G), P1(h["D"]), P1(h["E"]), P1(hg["T"]), P1(hg["A3"]),
                                                W1(hg["z1"], 8), W1(hg["z2"], 8), ok)])
    res["heg"] = bool(ok[0])
    a1, a2, z = out["ecddh"]
    orc.lib.orc_ecddh_verify(1, *[_p(a) for a in (P1(R.G), P1(e["h1"]), P1(e["g2"]), P1(e["h2"]), P1(a1), P1(a2), W1(z, 8), ok)])
    res["ecddh"] = bool(ok[0])
    orc.lib.orc_correct_key_verify(1, _p(W1(inp["ck"]["N"], 64)), _p(F.words(out["ck_sigma"], 64)), _p(ok))
    res["ck"] = bool(ok[0])
    x, y = out["cdlog"]
    orc.lib.orc_composite_dlog_verify(1, *[_p(a) for a in (W1(c["N"], 64), W1(c["g"], 64), W1(c["ni"], 64), W1(x, 64), W1(y, 73), ok)])
    res["cdlog"] = bool(ok[0])
    return res


def python_verdicts(inp, out):
    h, e, c = inp["heg"], inp["ecddh"], inp["cdlog"]
    return dict(dlog=R.dlog_verify(*out["dlog"]), pedersen=PG.pedersen_verify(out["pedersen"]),
                heg=PG.heg_verify(out["heg"], h["G"], R.H2, R.G, h["D"], h["E"]),
                ecddh=PG.ecddh_verify(R.G, e["h1"], e["g2"], e["h2"], *out["ecddh"]),
                ck=R.correct_key_verify(inp["ck"]["N"], out["ck_sigma"]),
                cdlog=R.composite_dlog_verify(c["N"], c["g"], c["ni"], *out["cdlog"]))


# ---- the HIP engine (only imported by the -m gpu tests) ------------------------------------------------------------------
def _d(ctx, arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


def _u(t):
    return t.cpu().numpy().view(np.uint32)


def engine_outputs(ctx, inp):
    """the same bundle through the C-ABI, under the context's profile (mpe_ctx_set_encoding)"""
    from multi_party_ecdsa_amd import engine as E
    d, p, h, e, c = inp["dlog"], inp["pedersen"], inp["heg"], inp["ecddh"], inp["cdlog"]
    col = lambda v, w: _d(ctx, W1(v, w))
    pts = lambda pt: _d(ctx, P1(pt))
    pk, Rr, z = E.dlog_prove(ctx, col(d["sk"], 8), col(d["nonce"], 8))
    pe = E.pedersen_prove(ctx, col(p["m"], 8), col(p["r"], 8), col(p["s1"], 8), col(p["s2"], 8))
    ped = dict(e=F.ints(_u(pe["e"]))[0], a1=F.points(_u(pe["a1"]))[0], a2=F.points(_u(pe["a2"]))[0], com=F.points(_u(pe["com"]))[0],
               z1=F.ints(_u(pe["z1"]))[0], z2=F.ints(_u(pe["z2"]))[0])
    hst = dict(G=pts(h["G"]), H=pts(R.H2), Y=pts(R.G), D=pts(h["D"]), E=pts(h["E"]))
    he = E.heg_prove(ctx, col(h["x"], 8), col(h["r"], 8), col(h["s1"], 8), col(h["s2"], 8), hst)
    heg = dict(T=F.points(_u(he["T"]))[0], A3=F.points(_u(he["A3"]))[0], z1=F.ints(_u(he["z1"]))[0], z2=F.ints(_u(he["z2"]))[0])
    dst = dict(g1=pts(R.G), h1=pts(e["h1"]), g2=pts(e["g2"]), h2=pts(e["h2"]))
    dd = E.ecddh_prove(ctx, col(e["x"], 8), col(e["s"], 8), dst)
    coms = [F.ints(_u(E.hash_commit_point(ctx, pts(x["point"]), col(x["blind"], 8))))[0] for x in inp["commit"]]
    sk = E.PaillierKeys(ctx, p=[inp["ck"]["p"]], q=[inp["ck"]["q"]])
    sigma = E.correct_key_prove(ctx, sk)
    cx, cy = E.composite_dlog_prove(ctx, col(c["N"], 64), col(c["g"], 64), col(c["ni"], 64), col(c["secret"], 64), col(c["r"], 16))
    ctx.sync()
    return dict(dlog=(F.points(_u(pk))[0], F.points(_u(Rr))[0], F.ints(_u(z))[0]), pedersen=ped, heg=heg,
                ecddh=(F.points(_u(dd["a1"]))[0], F.points(_u(dd["a2"]))[0], F.ints(_u(dd["z"]))[0]), commit=coms,
                ck_sigma=F.ints(_u(sigma).reshape(-1, 64)), cdlog=(F.ints(_u(cx))[0], F.ints(_u(cy))[0]))


def engine_verdicts(ctx, inp, out):
    from multi_party_ecdsa_amd import engine as E
    h, e, c = inp["heg"], inp["ecddh"], inp["cdlog"]
    col = lambda v, w: _d(ctx, W1(v, w))
    pts = lambda pt: _d(ctx, P1(pt))
    res = {}
    pk, Rr, z = out["dlog"]
    res["dlog"] = bool(E.dlog_verify(ctx, pts(pk), pts(Rr), col(z, 8)).cpu().numpy()[0])
    pe = out["pedersen"]
    proof = dict(com=pts(pe["com"]), e=col(pe["e"], 8), a1=pts(pe["a1"]), a2=pts(pe["a2"]), z1=col(pe["z1"], 8), z2=col(pe["z2"], 8))
    res["pedersen"] = bool(E.pedersen_verify(ctx, proof).cpu().numpy()[0])
    hg = out["heg"]
    hst = dict(G=pts(h["G"]), H=pts(R.H2), Y=pts(R.G), D=pts(h["D"]), E=pts(h["E"]))
    res["heg"] = bool(E.heg_verify(ctx, hst, dict(T=pts(hg["T"]), A3=pts(hg["A3"]), z1=col(hg["z1"], 8), z2=col(hg["z2"], 8))).cpu().numpy()[0])
    a1, a2, z = out["ecddh"]
    dst = dict(g1=pts(R.G), h1=pts(e["h1"]), g2=pts(e["g2"]), h2=pts(e["h2"]))
    res["ecddh"] = bool(E.ecddh_verify(ctx, dst, dict(a1=pts(a1), a2=pts(a2), z=col(z, 8))).cpu().numpy()[0])
    res["ck"] = bool(E.correct_key_verify(ctx, col(inp["ck"]["N"], 64), _d(ctx, F.words(out["ck_sigma"], 64))).cpu().numpy()[0])
    x, y = out["cdlog"]
    res["cdlog"] = bool(E.composite_dlog_verify(ctx, col(c["N"], 64), col(c["g"], 64), col(c["ni"], 64), col(x, 64), col(y, 73)).cpu().numpy()[0])
    return res
