"""Consumes tests/golden/ref_vectors.json — vectors dumped from the REAL crates by tools/rust_vectors/dump_vectors.rs — when a
maintainer has produced it (skipped otherwise: this image has no Rust toolchain).  Every crate-generated proof must be
accepted by the oracle's verifiers and every deterministic value reproduced byte for byte; this is what would turn the
"parity unpinned" of oracle/mpe_oracle.h into a pinned one.  Also: the wire module decodes its own encodings."""
import importlib.util
import json
import os

import numpy as np
import pytest

import enc_profiles as ENCS
import fixtures as F
import gg20_fixture as G
import orc
import pyref

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "ref_vectors.json")
spec = importlib.util.spec_from_file_location("mpe_wire", os.path.join(os.path.dirname(HERE), "multi_party_ecdsa_amd", "wire.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)           # by path: the package import needs the HIP library, this module does not

H = lambda s: int(s, 16)


def _pt(v):
    return (H(v["x"]), H(v["y"]))


SELFMADE = os.path.join(HERE, "golden", "selfmade_vectors.json")
SELFMADE_ALT = os.path.join(HERE, "golden", "selfmade_vectors_alt.json")


def diagnosed_profile(cases, what):
    """Self-diagnosing consumer: search every combination of the recalled conventions (enc_profiles.diagnose), say which one
    the file was produced under, and fail ONLY when some proof verifies under none."""
    prof, report = ENCS.diagnose(cases, wire=W)
    print(f"\n[{what}] encoding profile found: {prof!r}   (defaults of include/mpecdsa_hip.h: {prof == ENCS.DEFAULT})")
    for name in ("dlog", "pedersen", "heg", "ecddh", "correct_key", "composite_dlog"):
        if name in report:
            print(f"    {name:15s} combinations that verify: {report[name]['n_matches']}  first: {report[name]['matches'][:1]}")
    assert prof is not None, f"no combination of the known conventions verifies {report['no_combination_for']}: a recalled FORMULA differs, not an encoding"
    return prof


@pytest.mark.skipif(not os.path.exists(REF), reason="tests/golden/ref_vectors.json not produced yet (tools/rust_vectors/run.sh)")
def test_vectors_from_the_real_crates():
    """THE pin.  Whatever conventions the real crates use, the run reports them (-s shows the profile; it is also written to
    tests/golden/encoding_profile.json by tools/diagnose_encodings.py) and every crate-generated value must then be
    reproduced / accepted by the oracle configured with that profile."""
    doc = json.load(open(REF))
    assert "SELF-MADE" not in doc["crate"]
    prof = diagnosed_profile(doc["cases"], "ref_vectors.json")
    with ENCS.applied(prof):
        check_cases(doc["cases"])


def test_consumer_on_selfmade_vectors_of_the_same_schema():
    """the same checks on a file of the dump's schema produced by this repo's Python restatement (NOT a pin of the crates:
    it keeps the consumer exercised, so that the real dump is checked by code known to work)"""
    doc = json.load(open(SELFMADE))
    assert doc["schema"] == 1 and "SELF-MADE" in doc["crate"]
    prof = diagnosed_profile(doc["cases"], "selfmade_vectors.json")
    assert prof == ENCS.DEFAULT
    with ENCS.applied(prof):
        check_cases(doc["cases"])


def test_diagnoser_recovers_a_non_default_profile_and_the_oracle_follows_it():
    """selfmade_vectors_alt.json was written under the "all-alt" profile (compressed chain_point, empty zero, reversed
    transcript orders, big-endian mask blocks) in the OTHER wire style: the diagnoser must find exactly that profile from the
    vectors alone, the oracle configured with it must accept everything, and the oracle on its defaults must NOT."""
    doc = json.load(open(SELFMADE_ALT))
    assert doc["selfmade_profile"] == "all-alt"
    prof = diagnosed_profile(doc["cases"], "selfmade_vectors_alt.json")
    assert prof == ENCS.PROFILES["all-alt"]
    with ENCS.applied(prof):
        check_cases(doc["cases"])
    with pytest.raises(AssertionError):
        check_cases(doc["cases"])                    # defaults: the transcripts differ, so the proofs are rejected


def check_sampler(rec, N):
    """the sampling side (mpe_sample.h / oracle/sampler_oracle.c restate curv's `Samplable`): the byte -> integer rule on the dump's
    known strings must be the oracle's, and the crate's real draws must lie in the ranges the restatement samples from"""
    for kb in rec["known_bytes"]:
        buf, bits = bytes.fromhex(kb["bytes"]), kb["bits"]
        words = (bits + 31) // 32
        assert F.ints(orc.sample_rule(buf, bits, words))[0] == H(kb["value"]), ("BigInt::sample byte rule", bits)
    sb = rec["sample_bits"]
    draws = [H(v) for v in sb["draws"]]
    assert all(0 <= v < (1 << sb["bits"]) for v in draws) and max(v.bit_length() for v in draws) == sb["bits"]
    u = H(rec["sample_below"]["upper"])
    assert all(0 <= H(v) < u for v in rec["sample_below"]["draws"])
    lo, hi = H(rec["sample_range"]["lo"]), H(rec["sample_range"]["hi"])
    assert hi == N - 1 and all(lo <= H(v) < hi for v in rec["sample_range"]["draws"])
    assert all(0 < H(v) < pyref.Q for v in rec["scalar_random"])


def check_cases(cases):
    assert cases
    for c in cases:
        if "sampler" in c:
            check_sampler(c["sampler"], H(c["keys"]["N"]))
        k = c["keys"]
        N, p, q, Nt, h1, h2 = (H(k[f]) for f in ("N", "p", "q", "Nt", "h1", "h2"))
        assert p * q == N
        # Paillier
        pa = c["paillier"]
        assert pyref.paillier_encrypt(N, H(pa["m"]), H(pa["r"])) == H(pa["c"])
        assert F.ints(orc.paillier_decrypt(F.words([p], 32), F.words([q], 32), F.words([H(pa["c"])], 128)))[0] == H(pa["m"])
        # AliceProof: crate-generated, oracle-verified
        ap = {f: W.bigint_from_json(v) for f, v in c["alice_proof"]["proof"].items()}
        widths = dict(z=64, e=8, s=64, s1=25, s2=89)
        pr = {f: F.words([ap[f]], w) for f, w in widths.items()}
        ok = orc.alice_verify(F.words([N], 64), F.words([Nt], 64), F.words([h1], 64), F.words([h2], 64), None, None,
                              F.words([H(c["alice_proof"]["cipher"])], 128), pr)
        assert ok[0] == 1, "AliceProof of the crate rejected: transcript encoding differs"
        # PDL with slack
        pd = c["pdl"]
        pp = pd["proof"]
        prf = dict(z=F.words([W.bigint_from_json(pp["z"])], 64), u1=F.point_words([W.point_from_json(pp["u1"])]),
                   u2=F.words([W.bigint_from_json(pp["u2"])], 128), u3=F.words([W.bigint_from_json(pp["u3"])], 64),
                   s1=F.words([W.bigint_from_json(pp["s1"])], 25), s2=F.words([W.bigint_from_json(pp["s2"])], 64),
                   s3=F.words([W.bigint_from_json(pp["s3"])], 89))
        ok = orc.pdl_verify(F.words([N], 64), F.words([Nt], 64), F.words([h1], 64), F.words([h2], 64), None, None, F.words([H(pd["c"])], 128),
                            F.point_words([_pt(pd["Q"])]), F.point_words([_pt(pd["G"])]), prf)
        assert ok[0] == 1, "PDLwSlackProof of the crate rejected"
        assert W.point_from_json(pd["Q"]["serde"]) == _pt(pd["Q"])          # the serde form of a Point
        # DLogProof
        dl = c["dlog"]["proof"]
        ok = orc.dlog_verify(F.point_words([W.point_from_json(dl["pk"])]), F.point_words([W.point_from_json(dl["pk_t_rand_commitment"])]),
                             F.words([W.scalar_from_json(dl["challenge_response"])], 8))
        assert ok[0] == 1, "DLogProof of the crate rejected: chain_point / result_scalar differ"
        # PedersenProof, HomoELGamalProof, ECDDHProof
        pe = c["pedersen"]["proof"]
        okb = np.zeros(1, dtype=np.uint8)
        orc.lib.orc_pedersen_verify(1, *[orc._p(a) for a in (F.point_words([W.point_from_json(pe["com"])]), F.point_words([W.point_from_json(pe["a1"])]),
                                                             F.point_words([W.point_from_json(pe["a2"])]), F.words([W.scalar_from_json(pe["z1"])], 8),
                                                             F.words([W.scalar_from_json(pe["z2"])], 8), okb)])
        assert okb[0] == 1, "PedersenProof of the crate rejected"
        he, hp = c["heg"], c["heg"]["proof"]
        orc.lib.orc_heg_verify(1, *[orc._p(a) for a in (F.point_words([_pt(he["G"])]), F.point_words([pyref.H2]), F.point_words([pyref.G]),
                                                        F.point_words([_pt(he["D"])]), F.point_words([_pt(he["E"])]), F.point_words([W.point_from_json(hp["T"])]),
                                                        F.point_words([W.point_from_json(hp["A3"])]), F.words([W.scalar_from_json(hp["z1"])], 8),
                                                        F.words([W.scalar_from_json(hp["z2"])], 8), okb)])
        assert okb[0] == 1, "HomoELGamalProof of the crate rejected"
        dd, dp = c["ecddh"], c["ecddh"]["proof"]
        orc.lib.orc_ecddh_verify(1, *[orc._p(a) for a in (F.point_words([pyref.G]), F.point_words([_pt(dd["h1"])]), F.point_words([_pt(dd["g2"])]),
                                                          F.point_words([_pt(dd["h2"])]), F.point_words([W.point_from_json(dp["a1"])]),
                                                          F.point_words([W.point_from_json(dp["a2"])]), F.words([W.scalar_from_json(dp["z"])], 8), okb)])
        assert okb[0] == 1, "ECDDHProof of the crate rejected"
        # zk-paillier keygen proofs and kzen-paillier's Open (when the dump has them: schema additions of round 2)
        if "correct_key" in c:
            sig = [W.bigint_from_json(v) for v in c["correct_key"]["proof"]["sigma_vec"]]
            assert len(sig) == 11
            okb2 = np.zeros(1, dtype=np.uint8)
            orc.lib.orc_correct_key_verify(1, orc._p(F.words([N], 64)), orc._p(F.words(sig, 64)), orc._p(okb2))
            assert okb2[0] == 1, "NiCorrectKeyProof of the crate rejected: salt / mask generation differ"
        if "composite_dlog" in c and c["composite_dlog"]["verifies"]:
            cdp = c["composite_dlog"]["proof"]
            ni = H(c["composite_dlog"]["ni"]) if "ni" in c["composite_dlog"] else h2          # the crate dump proves the statement (N~, h1, h2) itself
            orc.lib.orc_composite_dlog_verify(1, orc._p(F.words([Nt], 64)), orc._p(F.words([h1], 64)), orc._p(F.words([ni], 64)),
                                              orc._p(F.words([W.bigint_from_json(cdp["x"])], 64)), orc._p(F.words([W.bigint_from_json(cdp["y"])], 73)), orc._p(okb))
            assert okb[0] == 1, "CompositeDLogProof of the crate rejected"
        if "open" in c:
            om, orr = orc.u32((1, 64)), orc.u32((1, 64))
            orc.lib.orc_paillier_open(1, 1, orc._p(F.words([p], 32)), orc._p(F.words([q], 32)), None, orc._p(F.words([H(c["open"]["c"])], 128)), orc._p(om), orc._p(orr))
            assert F.ints(om)[0] == H(c["open"]["m"]) and F.ints(orr)[0] == H(c["open"]["r"])
        # HashCommitment, base_point2
        hc = c["hash_commitment"]
        com = orc.u32((1, 8))
        orc.lib.orc_hash_commit_point(1, orc._p(F.point_words([_pt(hc["point"])])), orc._p(F.words([H(hc["blind"])], 8)), orc._p(com))
        assert F.ints(com)[0] == H(hc["com"])
        assert _pt(c["base_point2"]) == pyref.H2
        # MtA: MessageB of the crate passes verify_proofs_get_alpha's checks and gives the crate's alpha
        mt = c["mta"]
        mb = mt["m_b"]
        share = F.ints(orc.paillier_decrypt(F.words([p], 32), F.words([q], 32), F.words([W.bigint_from_json(mb["c"])], 128)))[0]
        assert share == H(mt["alice_share"]) and share % pyref.Q == H(mt["alpha"]["hex"])
        for prf_ in (mb["b_proof"], mb["beta_tag_proof"]):
            assert orc.dlog_verify(F.point_words([W.point_from_json(prf_["pk"])]), F.point_words([W.point_from_json(prf_["pk_t_rand_commitment"])]),
                                   F.words([W.scalar_from_json(prf_["challenge_response"])], 8))[0] == 1


@pytest.mark.parametrize("t,n,signers", [(1, 3, [0, 2]), (2, 4, [0, 1, 3])])
def test_wire_roundtrip_of_every_round_message(keys, t, n, signers):
    """record -> serde-shaped JSON `Msg<OfflineProtocolMessage>` -> record, for every message of a real session"""
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, 1, seed="wire")
    got = G.oracle_sign_ex(lk, nonces, 1)
    S = len(signers)
    for rnd in G.ROUNDS:
        for i in range(S):
            rec = got["slabs"][rnd][i, 0]
            msgs = W.record_to_msgs(rnd, rec, S, n, i + 1)
            assert len(msgs) == (S - 1 if rnd == 1 else 1)
            parsed = [json.loads(m) for m in msgs]
            assert all(m["sender"] == i + 1 for m in parsed)
            if rnd == 1:
                assert sorted(m["receiver"] for m in parsed) == [j + 1 for j in range(S) if j != i]       # P2P
            elif rnd != 7:
                assert parsed[0]["receiver"] is None and list(parsed[0]["body"]) == [f"M{G.ROUNDS.index(rnd) + 1}"]
            back = W.bodies_to_record(rnd, [(m["receiver"], m["body"]) for m in parsed], S, n, i + 1)
            assert np.array_equal(back, rec), (rnd, i)
    m0 = json.loads(W.record_to_msgs(0, got["slabs"][0][0, 0], S, n, 1)[0])["body"]["M1"]
    assert set(m0[0]) == {"c", "range_proofs"} and set(m0[0]["range_proofs"][0]) == {"z", "e", "s", "s1", "s2"} and set(m0[1]) == {"com"}


def test_primitive_decoders_accept_the_other_known_forms():
    x = 0x1234567890abcdef1234567890
    assert W.bigint_from_json(W.bigint_to_json(x)) == x
    assert W.bigint_from_json(list(x.to_bytes(13, "big"))) == x
    Pt = pyref.ec_mul(12345, pyref.G)
    unc = "04%064x%064x" % Pt
    assert W.point_from_json(W.point_to_json(Pt)) == Pt == W.point_from_json({"curve": "secp256k1", "point": unc})
    assert W.point_from_json({"curve": "secp256k1", "point": list(bytes.fromhex(unc))}) == Pt
    assert W.scalar_from_json(W.scalar_to_json(77)) == 77
    # a digit-only string is a decimal AND a hexadecimal numeral; the default reads curv's hex, strict refuses to guess
    assert W.bigint_from_json("10") == 16 and W.bigint_from_json("10", radix=10) == 10 and W.bigint_from_json("0a", strict=True) == 10
    with pytest.raises(ValueError, match="ambiguous"):
        W.bigint_from_json("1234", strict=True)
    assert W.bigint_from_json("0", strict=True) == 0
    with pytest.raises(ValueError):
        W.point_from_json({"curve": "secp256k1", "point": "02" + "%064x" % 5})          # x = 5 is not on the curve


def test_local_key_json_roundtrip_and_key_arrays(keys):
    """LocalKey (keygen/rounds.rs:311-322) as `gg20_keygen` stores it -> the arrays mpe_gg20_keys_create takes: identical to the
    fixture's for every party, own-secret rows only, and inconsistent shares are refused"""
    t, n = 2, 5
    lk = G.make_local_keys(keys, t, n, [0, 2, 4])
    A = lk["arrays"]
    xs, X, y = F.ints(A["x"]), F.points(A["X"]), F.points(A["y"])[0]
    Ns = [k.N for k in lk["keys"]]
    stm = [(k.Nt, k.h1, k.h2) for k in lk["keys"]]
    docs = [json.dumps(W.local_key_to_json(i + 1, t, n, lk["keys"][i].p, lk["keys"][i].q, xs[i], y, X, Ns, stm)) for i in range(n)]
    parsed = [W.local_key_from_json(d) for d in docs]
    assert [p_["i"] for p_ in parsed] == [1, 2, 3, 4, 5] and parsed[3]["x_i"] == xs[3]
    one = W.local_keys_to_arrays([parsed[2]])                                   # the deployment: one party per process
    assert one["own"] == [2] and (one["t"], one["n"]) == (t, n)
    for f in ("Nt", "h1", "h2", "y", "X"):
        assert np.array_equal(one["arrays"][f], A[f]), f
    assert np.array_equal(one["arrays"]["N"], F.words(Ns, 64))
    assert np.array_equal(one["arrays"]["x"][2], A["x"][2]) and not one["arrays"]["x"][[0, 1, 3, 4]].any()
    assert np.array_equal(one["arrays"]["p"][2], A["p"][2]) and not one["arrays"]["q"][[0, 1, 3, 4]].any()
    every = W.local_keys_to_arrays(parsed)                                      # the Simulation harness: all of them
    for f in ("x", "p", "q"):
        assert np.array_equal(every["arrays"][f], A[f]), f
    bad = json.loads(docs[1])
    bad["paillier_dk"]["p"] = W.bigint_to_json(lk["keys"][0].p)
    with pytest.raises(ValueError):
        W.local_key_from_json(bad)
    other = W.local_key_from_json(docs[0])
    other["y"] = X[0]
    with pytest.raises(ValueError):
        W.local_keys_to_arrays([parsed[1], other])


STYLES = [W.DEFAULT_STYLE, W.HEX_STYLE, W.Style(point="bytes", scalar="hex", bigint="bytes", paillier="decimal"),
          W.Style(point="hex", scalar="bytes", bigint="hex", paillier="decimal")]


@pytest.mark.parametrize("style", STYLES, ids=lambda s: f"{s.point}-{s.scalar}-{s.bigint}-{s.paillier}")
def test_every_encoder_style_decodes_to_the_same_records_and_keys(keys, style):
    """both believed serde forms of every primitive (byte arrays vs hex strings for Point / Scalar, decimal vs hex for the
    kzen-paillier keys): whatever the encoder writes, the decoders read the same values back"""
    t, n, signers = 1, 3, [0, 2]
    lk = G.make_local_keys(keys, t, n, signers)
    got = G.oracle_sign_ex(lk, G.make_nonces(lk, 1, seed="wire-style"), 1)
    S = len(signers)
    for rnd in G.ROUNDS:
        for i in range(S):
            rec = got["slabs"][rnd][i, 0]
            parsed = [json.loads(m) for m in W.record_to_msgs(rnd, rec, S, n, i + 1, style=style)]
            assert np.array_equal(W.bodies_to_record(rnd, [(m["receiver"], m["body"]) for m in parsed], S, n, i + 1), rec), (rnd, i)
    body = json.loads(W.record_to_msgs(3, got["slabs"][3][0, 0], S, n, 1, style=style)[0])["body"]["M4"]
    assert isinstance(body["g_gamma_i"]["point"], list if style.point == "bytes" else str)
    A = lk["arrays"]
    xs, X, y = F.ints(A["x"]), F.points(A["X"]), F.points(A["y"])[0]
    Ns, stm = [k.N for k in lk["keys"]], [(k.Nt, k.h1, k.h2) for k in lk["keys"]]
    for i in range(n):
        doc = W.local_key_to_json(i + 1, t, n, lk["keys"][i].p, lk["keys"][i].q, xs[i], y, X, Ns, stm, style=style)
        if style.paillier == "decimal":
            assert doc["paillier_dk"]["p"].isdigit() and doc["paillier_key_vec"][0]["n"] == str(Ns[0])
        back = W.local_key_from_json(json.dumps(doc))
        assert (back["p"], back["q"], back["x_i"], back["y"], back["pk_vec"], back["N_vec"], back["stm_vec"]) == \
               (lk["keys"][i].p, lk["keys"][i].q, xs[i], y, X, Ns, stm)


def test_digit_only_decimal_key_fields_are_not_misread_as_hex(keys):
    """kzen-paillier writes radix-10 strings; a digit-only string also parses as hex.  The decoder must pick the
    reading in which p * q == n — for decimal documents AND for hex documents whose digits happen to be all 0-9."""
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    A = lk["arrays"]
    xs, X, y = F.ints(A["x"]), F.points(A["X"]), F.points(A["y"])[0]
    Ns, stm = [k.N for k in lk["keys"]], [(k.Nt, k.h1, k.h2) for k in lk["keys"]]
    doc = W.local_key_to_json(2, 1, 3, lk["keys"][1].p, lk["keys"][1].q, xs[1], y, X, Ns, stm)
    assert int(doc["paillier_dk"]["p"]) == lk["keys"][1].p and int(doc["paillier_dk"]["p"], 16) != lk["keys"][1].p
    doc["some_future_field"] = {"ignored": True}                                      # unknown fields are tolerated
    back = W.local_key_from_json(doc)
    assert back["p"] == lk["keys"][1].p and back["N_vec"] == Ns
    assert W.paillier_bigint_readings("255") == [255, 0x255] and W.paillier_bigint_readings("ff") == [255]
    # a toy hex document whose digits are all decimal digits: 0x11 * 0x13 = 0x143
    toy = dict(doc, paillier_dk={"p": "11", "q": "13"}, paillier_key_vec=[{"n": "99"}, {"n": "143"}, {"n": "77"}])
    toy_back = W.local_key_from_json(toy)
    assert (toy_back["p"], toy_back["q"], toy_back["N_vec"][1]) in ((11, 13, 143), (0x11, 0x13, 0x143))
    bad = dict(doc, paillier_dk={"p": doc["paillier_dk"]["q"], "q": doc["paillier_dk"]["q"]})
    with pytest.raises(ValueError):
        W.local_key_from_json(bad)


def test_local_share_json_shaped_like_the_reference_writes_it(keys):
    """A document with the layout `gg20_keygen` leaves on disk (examples/gg20_keygen.rs:52-56: serde_json of
    LocalKey<Secp256k1>, keygen/rounds.rs:311-322) in the believed curv 0.9 / kzen-paillier forms — byte arrays for points and
    scalars, decimal strings for the Paillier key, hex strings for N~, h1, h2 — becomes the key arrays of the engine."""
    with open(os.path.join(HERE, "golden", "local_share_like.json")) as f:
        doc = json.load(f)
    assert isinstance(doc["pk_vec"][0]["point"], list) and doc["paillier_dk"]["p"].isdigit() and isinstance(doc["h1_h2_n_tilde_vec"][0]["N"], str)
    lk = W.local_key_from_json(doc)
    assert lk["i"] == doc["i"] and lk["p"] * lk["q"] == lk["N_vec"][lk["i"] - 1]
    arr = W.local_keys_to_arrays([lk])
    assert arr["own"] == [lk["i"] - 1] and arr["arrays"]["X"].shape == (lk["n"], 16)
    want = G.make_local_keys(keys, 1, 3, [0, 1])["arrays"]
    for f in ("Nt", "h1", "h2", "y", "X"):
        assert np.array_equal(arr["arrays"][f], want[f]), f
