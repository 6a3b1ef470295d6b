"""Generates tests/golden/selfmade_vectors.json: a file with EXACTLY the schema tools/rust_vectors/dump_vectors.rs prints, but
produced by this repository's own pure-Python restatement (tests/pyref.py, tests/pyref_gg20.py) and encoded with
multi_party_ecdsa_amd/wire.py's default serde forms.  It is NOT a pin of the Rust crates — it exists so that the consumers of
the real dump (tests/test_ref_vectors_cpu.py for the oracle, tests/test_ref_vectors_gpu.py for the HIP engine) are exercised on
every run and are known to work the day tests/golden/ref_vectors.json is produced (tools/rust_vectors/run.sh).
Two files: selfmade_vectors.json under the default encoding profile and wire style, and selfmade_vectors_alt.json under the
"all-alt" profile of tests/enc_profiles.py (compressed chain_point, empty zero, reversed transcript orders, big-endian mask) in
the hex wire style — so the consumers' diagnoser (enc_profiles.diagnose) is known to RECOVER a non-default profile from vectors.
Run from the repo root:  python tests/golden/make_selfmade_vectors.py"""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("mpe_wire", os.path.join(ROOT, "multi_party_ecdsa_amd", "wire.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)
import fixtures as F      # noqa: E402
import pyref as R         # noqa: E402
import pyref_gg20 as PG   # noqa: E402

hx = lambda x: "%x" % x


import enc_profiles as ENCS   # noqa: E402

STYLE = W.DEFAULT_STYLE


class _Styled:
    """wire.py's encoders with the style of the file being written"""
    point_to_json = staticmethod(lambda p: W.point_to_json(p, STYLE))
    scalar_to_json = staticmethod(lambda x: W.scalar_to_json(x, STYLE))
    bigint_to_json = staticmethod(lambda x: W.bigint_to_json(x, STYLE))


def pt(p):
    return {"x": hx(p[0]), "y": hx(p[1]), "bytes_compressed": R.pt_bytes(p, True).hex(), "serde": W.point_to_json(p, STYLE)}


def sc(s):
    return {"hex": hx(s), "serde": W.scalar_to_json(s, STYLE)}


def dlog_json(pk, Rr, z):
    return {"pk": W.point_to_json(pk, STYLE), "pk_t_rand_commitment": W.point_to_json(Rr, STYLE), "challenge_response": W.scalar_to_json(z, STYLE)}


def sampler_record(r, N):
    """the same record tools/rust_vectors/dump_vectors.rs emits: the byte rule of BigInt::sample on known strings + draws in range"""
    known = []
    for ln, bits in ((32, 256), (1, 7), (1, 1), (2, 9), (5, 33), (256, 2047), (256, 2048), (257, 2050), (352, 2816)):
        buf = bytes((((i * 167 + 13 + ln) & 0xFF) | (0x80 if i == 0 else 0)) for i in range(ln))
        known.append({"bytes": buf.hex(), "bits": bits, "value": hx(int.from_bytes(buf, "big") >> (ln * 8 - bits))})
    u = (1 << 300) + 12345
    return {"known_bytes": known,
            "sample_bits": {"bits": 7, "draws": [hx(r.bits(7)) for _ in range(64)]},
            "sample_below": {"upper": hx(u), "draws": [hx(r.below(u)) for _ in range(64)]},
            "sample_range": {"lo": "1", "hi": hx(N - 1), "draws": [hx(1 + r.below(N - 2)) for _ in range(16)]},
            "scalar_random": [hx(1 + r.below(R.Q - 1)) for _ in range(16)]}


def main():
    global STYLE
    for profile, style, fname, ncases, seed in (("default", W.DEFAULT_STYLE, "selfmade_vectors.json", 4, "selfmade-vectors-v1"),
                                                ("all-alt", W.HEX_STYLE, "selfmade_vectors_alt.json", 2, "selfmade-vectors-alt-v1")):
        STYLE = style
        with R.use_encoding(ENCS.PROFILES[profile]):
            write_file(profile, fname, ncases, seed)


def write_file(profile, fname, ncases, seed):
    keys = F.load_keys()
    r = F.Rng(seed)
    rs = lambda: r.below(R.Q - 1) + 1
    cases = []
    Wd = _Styled
    for i in range(ncases):
        ek, st = keys[i], keys[8 + i]
        N, NN = ek.N, ek.N * ek.N
        a, b, l = rs(), rs(), rs()
        rr = r.coprime_below(N)
        c = R.paillier_encrypt(N, a, rr)
        an = F.alice_nonces(r, ek, st)
        alice = R.alice_generate(N, st.Nt, st.h1, st.h2, a, c, rr, **an)
        assert R.alice_verify(N, st.Nt, st.h1, st.h2, c, alice)
        # MtA
        beta_tag, r_b = r.below(N), r.coprime_below(N)
        c_b = pow(c, b, NN) * R.paillier_encrypt(N, beta_tag, r_b) % NN
        bpk, bR, bz = R.dlog_prove(b, rs())
        tpk, tR, tz = R.dlog_prove(beta_tag % R.Q, rs())
        share = R.paillier_decrypt_textbook(ek.p, ek.q, c_b)
        alpha, beta = share % R.Q, (-beta_tag) % R.Q
        assert (alpha + beta) % R.Q == a * b % R.Q
        Rp = R.ec_mul(rs(), R.G)
        Qp = R.ec_mul(a, Rp)
        pn = F.pdl_nonces(r, ek, st)
        pdl = R.pdl_prove(N, st.Nt, st.h1, st.h2, c, Qp, Rp, a, rr, **pn)
        assert R.pdl_verify(N, st.Nt, st.h1, st.h2, c, Qp, Rp, pdl)
        dpk, dR, dz = R.dlog_prove(a, rs())
        ped = PG.pedersen_prove(a, l, rs(), rs())
        T, S = ped["com"], R.ec_mul(a, Rp)
        heg = PG.heg_prove(l, a, rs(), rs(), Rp, R.H2, R.G, T, S)
        assert PG.heg_verify(heg, Rp, R.H2, R.G, T, S)
        h1p = R.ec_mul(a, R.G)
        a1, a2, zz = PG.ecddh_prove(a, rs(), R.G, h1p, Rp, S)
        blind = r.bits(256)
        g_gamma = R.ec_mul(b, R.G)
        com = PG.hash_commitment(g_gamma, blind)
        sigma = R.correct_key_prove(ek.p, ek.q)
        secret = r.below(st.Nt >> 2)
        ni = pow(pow(st.h1, secret, st.Nt), -1, st.Nt)          # a statement of its own (the fixture's h2 has an unknown exponent)
        cdx, cdy = R.composite_dlog_prove(st.Nt, st.h1, ni, secret, r.bits(512))
        om, orr = PG.paillier_open(ek.p, ek.q, c)
        big = lambda d: {k: Wd.bigint_to_json(v) for k, v in d.items()}
        pdl_j = big({k: v for k, v in pdl.items() if k != "u1"})
        pdl_j["u1"] = Wd.point_to_json(pdl["u1"])
        cases.append({
            "keys": {"N": hx(N), "p": hx(ek.p), "q": hx(ek.q), "Nt": hx(st.Nt), "h1": hx(st.h1), "h2": hx(st.h2)},
            "paillier": {"m": hx(a), "r": hx(rr), "c": hx(c)},
            "alice_proof": {"a": hx(a), "cipher": hx(c), "proof": big(alice)},
            "mta": {"a": sc(a), "b": sc(b), "m_a": {"c": Wd.bigint_to_json(c), "range_proofs": [big(alice)]}, "m_a_randomness": hx(rr),
                    "m_b": {"c": Wd.bigint_to_json(c_b), "b_proof": dlog_json(bpk, bR, bz), "beta_tag_proof": dlog_json(tpk, tR, tz)},
                    "beta": sc(beta), "beta_randomness": hx(r_b), "beta_tag": hx(beta_tag), "alpha": sc(alpha), "alice_share": hx(share)},
            "pdl": {"x": sc(a), "r": hx(rr), "c": hx(c), "Q": pt(Qp), "G": pt(Rp), "proof": pdl_j},
            "dlog": {"sk": sc(a), "proof": dlog_json(dpk, dR, dz), "pk": pt(dpk)},
            "pedersen": {"m": sc(a), "r": sc(l), "com": pt(T),
                         "proof": {"e": Wd.scalar_to_json(ped["e"]), "a1": Wd.point_to_json(ped["a1"]), "a2": Wd.point_to_json(ped["a2"]),
                                   "com": Wd.point_to_json(ped["com"]), "z1": Wd.scalar_to_json(ped["z1"]), "z2": Wd.scalar_to_json(ped["z2"])}},
            "heg": {"x": sc(l), "r": sc(a), "G": pt(Rp), "D": pt(T), "E": pt(S),
                    "proof": {"T": Wd.point_to_json(heg["T"]), "A3": Wd.point_to_json(heg["A3"]), "z1": Wd.scalar_to_json(heg["z1"]),
                              "z2": Wd.scalar_to_json(heg["z2"])}},
            "ecddh": {"x": sc(a), "g2": pt(Rp), "h1": pt(h1p), "h2": pt(S),
                      "proof": {"a1": Wd.point_to_json(a1), "a2": Wd.point_to_json(a2), "z": Wd.scalar_to_json(zz)}},
            "correct_key": {"proof": {"sigma_vec": [Wd.bigint_to_json(v) for v in sigma]}},
            "composite_dlog": {"secret": hx(secret), "ni": hx(ni), "proof": {"x": Wd.bigint_to_json(cdx), "y": Wd.bigint_to_json(cdy)}, "verifies": True},
            "open": {"c": hx(c), "m": hx(om), "r": hx(orr)},
            "hash_commitment": {"point": pt(g_gamma), "blind": hx(blind), "com": hx(com)},
            "base_point2": pt(R.H2),
            "sampler": sampler_record(r, N),
        })
    doc = {"schema": 1, "crate": "SELF-MADE (tests/golden/make_selfmade_vectors.py over tests/pyref*.py) - NOT multi-party-ecdsa / curv / kzen-paillier",
           "selfmade_profile": profile, "cases": cases}
    with open(os.path.join(HERE, fname), "w") as f:
        json.dump(doc, f)
    print(fname, profile, len(cases), "cases")


if __name__ == "__main__":
    main()
