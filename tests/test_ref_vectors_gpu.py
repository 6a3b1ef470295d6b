"""The HIP engine against vectors of tools/rust_vectors/dump_vectors.rs's schema: tests/golden/ref_vectors.json (from the REAL
crates, produced by tools/rust_vectors/run.sh on a machine with cargo — skipped while it does not exist) and
tests/golden/selfmade_vectors.json (same schema, produced by this repo's Python restatement: keeps this consumer exercised).
Every proof in the file must be ACCEPTED by the engine's verifiers (`mpe_alice_verify`, `mpe_pdl_verify`, `mpe_dlog_verify`,
`mpe_pedersen_verify`, `mpe_heg_verify`, `mpe_ecddh_verify`, `mpe_correct_key_verify`, `mpe_composite_dlog_verify`,
`mpe_mta_verify_get_alpha`) — a Fiat-Shamir verifier only accepts when it rebuilds the prover's transcript byte for byte — and
every deterministic value (Paillier ciphertext / plaintext, `Paillier::open`, the hash commitment, alpha) reproduced."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

import fixtures as F
import pyref

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "ref_vectors.json")
SELFMADE = os.path.join(HERE, "golden", "selfmade_vectors.json")
SELFMADE_ALT = os.path.join(HERE, "golden", "selfmade_vectors_alt.json")
spec = importlib.util.spec_from_file_location("mpe_wire", os.path.join(os.path.dirname(HERE), "multi_party_ecdsa_amd", "wire.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)
H = lambda s: int(s, 16)
_pt = lambda v: (H(v["x"]), H(v["y"]))


def check_cases_on_gpu(ctx, cases):
    from multi_party_ecdsa_amd import engine as E
    B = len(cases)
    dv = lambda vals, w: torch.from_numpy(F.words(vals, w).view(np.int32)).to(ctx.device)
    dp = lambda pts: torch.from_numpy(F.point_words(pts).view(np.int32)).to(ctx.device)
    u32 = lambda t: t.cpu().numpy().view(np.uint32)
    K = [{f: H(c["keys"][f]) for f in ("N", "p", "q", "Nt", "h1", "h2")} for c in cases]
    sk = E.PaillierKeys(ctx, p=[k["p"] for k in K], q=[k["q"] for k in K])
    pk = E.PaillierKeys(ctx, N=[k["N"] for k in K])
    stm = E.Statements(ctx, [k["Nt"] for k in K], [k["h1"] for k in K], [k["h2"] for k in K])
    idx = torch.arange(B, dtype=torch.int32, device=ctx.device)
    ones = [1] * B
    # Paillier: the crate's ciphertext from (m, r), by the holder and by a peer; decrypt; open
    m, r, c = ([H(x["paillier"][f]) for x in cases] for f in ("m", "r", "c"))
    assert F.ints(u32(sk.encrypt_device(dv(m, 64), dv(r, 64), idx))) == c == F.ints(u32(pk.encrypt_device(dv(m, 64), dv(r, 64), idx)))
    assert F.ints(u32(sk.decrypt_device(dv(c, 128), idx))) == m
    if all("open" in x for x in cases):
        om, orr = E.paillier_open(ctx, sk, dv([H(x["open"]["c"]) for x in cases], 128), idx)
        assert F.ints(u32(om)) == [H(x["open"]["m"]) for x in cases] and F.ints(u32(orr)) == [H(x["open"]["r"]) for x in cases]
    # AliceProof
    widths = dict(z=64, e=8, s=64, s1=25, s2=89)
    pr = {f: dv([W.bigint_from_json(x["alice_proof"]["proof"][f]) for x in cases], w) for f, w in widths.items()}
    ok = E.alice_verify(ctx, pk, stm, dv([H(x["alice_proof"]["cipher"]) for x in cases], 128), pr, idx, idx)
    assert list(ok.cpu().numpy()) == ones, "AliceProof of the file rejected by mpe_alice_verify"
    # PDL with slack
    pw = dict(z=64, u2=128, u3=64, s1=25, s2=64, s3=89)
    pp = {f: dv([W.bigint_from_json(x["pdl"]["proof"][f]) for x in cases], w) for f, w in pw.items()}
    pp["u1"] = dp([W.point_from_json(x["pdl"]["proof"]["u1"]) for x in cases])
    ok = E.pdl_verify(ctx, pk, stm, dv([H(x["pdl"]["c"]) for x in cases], 128), dp([_pt(x["pdl"]["Q"]) for x in cases]),
                      dp([_pt(x["pdl"]["G"]) for x in cases]), pp, idx, idx)
    assert list(ok.cpu().numpy()) == ones, "PDLwSlackProof of the file rejected by mpe_pdl_verify"
    # DLogProof
    dl = [x["dlog"]["proof"] for x in cases]
    ok = E.dlog_verify(ctx, dp([W.point_from_json(d["pk"]) for d in dl]), dp([W.point_from_json(d["pk_t_rand_commitment"]) for d in dl]),
                       dv([W.scalar_from_json(d["challenge_response"]) for d in dl], 8))
    assert list(ok.cpu().numpy()) == ones, "DLogProof of the file rejected by mpe_dlog_verify"
    # Pedersen, HomoELGamal, ECDDH
    pe = [x["pedersen"]["proof"] for x in cases]
    ok = E.pedersen_verify(ctx, dict(com=dp([W.point_from_json(d["com"]) for d in pe]), e=dv([W.scalar_from_json(d["e"]) for d in pe], 8),
                                     a1=dp([W.point_from_json(d["a1"]) for d in pe]), a2=dp([W.point_from_json(d["a2"]) for d in pe]),
                                     z1=dv([W.scalar_from_json(d["z1"]) for d in pe], 8), z2=dv([W.scalar_from_json(d["z2"]) for d in pe], 8)))
    assert list(ok.cpu().numpy()) == ones, "PedersenProof of the file rejected by mpe_pedersen_verify"
    he = [x["heg"] for x in cases]
    stt = dict(G=dp([_pt(d["G"]) for d in he]), H=dp([pyref.H2] * B), Y=dp([pyref.G] * B), D=dp([_pt(d["D"]) for d in he]), E=dp([_pt(d["E"]) for d in he]))
    ok = E.heg_verify(ctx, stt, dict(T=dp([W.point_from_json(d["proof"]["T"]) for d in he]), A3=dp([W.point_from_json(d["proof"]["A3"]) for d in he]),
                                     z1=dv([W.scalar_from_json(d["proof"]["z1"]) for d in he], 8), z2=dv([W.scalar_from_json(d["proof"]["z2"]) for d in he], 8)))
    assert list(ok.cpu().numpy()) == ones, "HomoELGamalProof of the file rejected by mpe_heg_verify"
    dd = [x["ecddh"] for x in cases]
    stt = dict(g1=dp([pyref.G] * B), h1=dp([_pt(d["h1"]) for d in dd]), g2=dp([_pt(d["g2"]) for d in dd]), h2=dp([_pt(d["h2"]) for d in dd]))
    ok = E.ecddh_verify(ctx, stt, dict(a1=dp([W.point_from_json(d["proof"]["a1"]) for d in dd]), a2=dp([W.point_from_json(d["proof"]["a2"]) for d in dd]),
                                       z=dv([W.scalar_from_json(d["proof"]["z"]) for d in dd], 8)))
    assert list(ok.cpu().numpy()) == ones, "ECDDHProof of the file rejected by mpe_ecddh_verify"
    # keygen proofs
    if all("correct_key" in x for x in cases):
        sig = torch.stack([dv([W.bigint_from_json(v) for v in x["correct_key"]["proof"]["sigma_vec"]], 64) for x in cases])
        ok = E.correct_key_verify(ctx, dv([k["N"] for k in K], 64), sig.contiguous())
        assert list(ok.cpu().numpy()) == ones, "NiCorrectKeyProof of the file rejected by mpe_correct_key_verify"
    cd = [x for x in cases if "composite_dlog" in x and x["composite_dlog"]["verifies"]]
    if cd:
        kk = [{f: H(x["keys"][f]) for f in ("Nt", "h1", "h2")} for x in cd]
        ni = [H(x["composite_dlog"]["ni"]) if "ni" in x["composite_dlog"] else k_["h2"] for x, k_ in zip(cd, kk)]
        ok = E.composite_dlog_verify(ctx, dv([k_["Nt"] for k_ in kk], 64), dv([k_["h1"] for k_ in kk], 64), dv(ni, 64),
                                     dv([W.bigint_from_json(x["composite_dlog"]["proof"]["x"]) for x in cd], 64),
                                     dv([W.bigint_from_json(x["composite_dlog"]["proof"]["y"]) for x in cd], 73))
        assert list(ok.cpu().numpy()) == [1] * len(cd), "CompositeDLogProof of the file rejected by mpe_composite_dlog_verify"
    # HashCommitment
    hc = [x["hash_commitment"] for x in cases]
    com = E.hash_commit_point(ctx, dp([_pt(d["point"]) for d in hc]), dv([H(d["blind"]) for d in hc], 8))
    assert F.ints(u32(com)) == [H(d["com"]) for d in hc]
    assert all(_pt(x["base_point2"]) == pyref.H2 for x in cases)
    # MtA: MessageB of the file through verify_proofs_get_alpha (decrypt, both DLog proofs, g^alice_share = b*g^a... check)
    mt = [x["mta"] for x in cases]
    dl_ = lambda which: dict(pk=dp([W.point_from_json(d["m_b"][which]["pk"]) for d in mt]),
                             R=dp([W.point_from_json(d["m_b"][which]["pk_t_rand_commitment"]) for d in mt]),
                             z=dv([W.scalar_from_json(d["m_b"][which]["challenge_response"]) for d in mt], 8))
    alpha, share, ok = E.mta_verify_get_alpha(ctx, sk, dv([W.bigint_from_json(d["m_b"]["c"]) for d in mt], 128), dl_("b_proof"), dl_("beta_tag_proof"),
                                              dv([H(d["a"]["hex"]) for d in mt], 8), idx)
    assert list(ok.cpu().numpy()) == ones, "MessageB of the file rejected by mpe_mta_verify_get_alpha"
    assert F.ints(u32(alpha)) == [H(d["alpha"]["hex"]) for d in mt] and F.ints(u32(share)) == [H(d["alice_share"]) for d in mt]
    ctx.sync()


def _ctx_for(cases, what):
    """a context configured with the profile the vectors were produced under (found by enc_profiles.diagnose from the vectors
    alone: every point form x permutation x zero encoding x mask order) — the engine never needs a rebuild to follow the crates"""
    import enc_profiles as ENCS
    from multi_party_ecdsa_amd import engine as E
    prof, report = ENCS.diagnose(cases, wire=W)
    print(f"\n[{what}] encoding profile found: {prof!r}")
    assert prof is not None, f"no combination of the known conventions verifies {report['no_combination_for']}"
    return E.Context(0, encoding=prof.as_dict()), prof


@pytest.mark.skipif(not os.path.exists(REF), reason="tests/golden/ref_vectors.json not produced yet (tools/rust_vectors/run.sh)")
def test_engine_accepts_the_vectors_of_the_real_crates():
    doc = json.load(open(REF))
    assert "SELF-MADE" not in doc["crate"]
    ctx, _ = _ctx_for(doc["cases"], "ref_vectors.json")
    check_cases_on_gpu(ctx, doc["cases"])


def test_engine_consumer_on_selfmade_vectors_of_the_same_schema(gpu_ctx):
    import enc_profiles as ENCS
    doc = json.load(open(SELFMADE))
    assert doc["schema"] == 1 and "SELF-MADE" in doc["crate"]
    ctx, prof = _ctx_for(doc["cases"], "selfmade_vectors.json")
    assert prof == ENCS.DEFAULT
    check_cases_on_gpu(ctx, doc["cases"])
    check_cases_on_gpu(gpu_ctx, doc["cases"])                 # a context left on its defaults is the same thing


def test_engine_follows_a_non_default_profile_found_from_the_vectors(gpu_ctx):
    """selfmade_vectors_alt.json: written under the "all-alt" profile in the hex wire style.  The diagnoser recovers the profile,
    a context configured with it accepts every proof and reproduces every value; the default context must reject the
    sigma proofs (their transcripts differ) — proof that the switch reaches the kernels."""
    import enc_profiles as ENCS
    doc = json.load(open(SELFMADE_ALT))
    ctx, prof = _ctx_for(doc["cases"], "selfmade_vectors_alt.json")
    assert prof == ENCS.PROFILES["all-alt"] and ctx.encoding() == prof.as_dict()
    check_cases_on_gpu(ctx, doc["cases"])
    with pytest.raises(AssertionError, match="rejected"):
        check_cases_on_gpu(gpu_ctx, doc["cases"])
