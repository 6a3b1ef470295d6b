"""The recalled byte-level conventions of the un-vendored crates as PROFILES, and the diagnoser that recovers the profile a
vector file was produced under.

The reference's Fiat-Shamir transcripts are built by curv-kzen 0.9 / zk-paillier 0.4.3 (Cargo.toml:36-47), which are not under
/root/reference: how `DigestExt::chain_point` serialises a point, what `BigInt::to_bytes()` gives for zero, the order of the
points inside each sigma proof's challenge, zk-paillier's salt and mask generation are RECALLED.  The HIP engine
(`mpe_encoding`, include/mpecdsa_hip.h), the C oracle (`orc_encoding`) and the Python restatement (`pyref.Encoding`) all take
them as a run-time profile with the same fields.  This module
  * names the alternatives the test-suite exercises (`PROFILES`) — every alternative is checked GPU == oracle == Python;
  * switches the two CPU restatements together (`applied`);
  * `diagnose(cases)`: given cases in the schema of tools/rust_vectors/dump_vectors.rs, searches EVERY combination (both point
    forms x every permutation of every proof's point list, both zero encodings, both mask orders, the CompositeDLogProof
    field orders) with nothing but hashlib and Python integers, and reports which one makes each crate-generated proof
    verify.  A disagreement between the real crates and this repository's defaults is then a profile to install
    (`mpe_ctx_set_encoding`), not a kernel to rewrite.
Test infrastructure: nothing here is imported by the product."""
import hashlib
import itertools
import json
import os

import pyref as R

HERE = os.path.dirname(os.path.abspath(__file__))

DEFAULT = R.Encoding()
# Every alternative of every field appears in at least one profile; "all-alt" turns everything at once (the permutations are
# arbitrary derangements: nothing may depend on a particular order).
PROFILES = {
    "default": DEFAULT,
    "compressed": DEFAULT.replace(chain_point=1),
    "zero-empty": DEFAULT.replace(zero_bytes=1),
    "mask-be": DEFAULT.replace(ck_mask_order=1, ck_salt=0x5A656E4B),
    "reordered": DEFAULT.replace(ord_dlog=(2, 0, 1), ord_pedersen=(2, 3, 4, 0, 1), ord_heg=(2, 3, 4, 5, 6, 0, 1),
                                 ord_ecddh=(4, 5, 0, 2, 1, 3), ord_cdlog=(2, 1, 3, 0)),
    "all-alt": R.Encoding(chain_point=1, zero_bytes=1, ck_mask_order=1, ck_salt=0x4B5A656E, ord_dlog=(1, 2, 0),
                          ord_pedersen=(4, 3, 2, 1, 0), ord_heg=(6, 5, 4, 3, 2, 1, 0), ord_ecddh=(1, 0, 3, 2, 5, 4), ord_cdlog=(3, 2, 1, 0)),
}
ALT_NAMES = [k for k in PROFILES if k != "default"]


class applied:
    """with applied(enc): ...  — the Python restatement AND the C oracle hash under `enc`; both are restored afterwards."""

    def __init__(self, enc):
        self.enc = enc

    def __enter__(self):
        import orc
        self.orc = orc
        self.prev = orc.get_encoding()
        orc.set_encoding(self.enc.as_dict())
        self.ctxmgr = R.use_encoding(self.enc)
        self.ctxmgr.__enter__()
        return self.enc

    def __exit__(self, *a):
        self.ctxmgr.__exit__(*a)
        self.orc.set_encoding(self.prev)


def load_profile(path):
    """a profile written by tools/diagnose_encodings.py"""
    with open(path) as f:
        return R.Encoding(**json.load(f)["profile"])


# ---- the diagnoser ---------------------------------------------------------------------------------------------------
def _challenge(points, order, compressed):
    h = hashlib.sha256()
    for k in order:
        h.update(R.pt_bytes(points[k], compressed))
    return int.from_bytes(h.digest(), "big") % R.Q


def _mul_many(es, P):
    """[e P for e in es]: the C oracle's batch multiplication when it is loadable (0.15 ms each), else Python (5 ms each)"""
    try:
        import fixtures as F
        import orc
        return F.points(orc.ec_mul(F.words(es, 8), F.point_words([P] * len(es))))
    except Exception:
        return [R.ec_mul(e, P) for e in es]


def _search_points(points, base, target, confirm=None):
    """every (chain_point, order) whose challenge e satisfies e * base == target (and confirm(e), checked on the survivors)"""
    cands = [(cp, order) for cp in (0, 1) for order in itertools.permutations(range(len(points)))]
    es = [_challenge(points, order, bool(cp)) for cp, order in cands]
    got = _mul_many(es, base)
    return [c for c, e, g in zip(cands, es, got) if g == target and (confirm is None or confirm(e))]


def _sub(a, b):
    return R.ec_add(a, R.ec_neg(b))


def diagnose(cases, wire=None):
    """cases: list in the dump's schema; wire: the decoder module (multi_party_ecdsa_amd/wire.py loaded by path).
    Returns (profile, report).  `report[proof]` lists what was found; profile is None when some proof verifies under NO
    combination (then the recalled *formulas*, not only the encodings, differ — report says which proof)."""
    c = cases[0]
    W = wire
    H = lambda s: int(s, 16)
    pt = lambda v: (H(v["x"]), H(v["y"]))
    report, prof, unique = {}, {}, True

    def settle(name, hits, identity_len):
        nonlocal unique
        report[name] = {"matches": [{"chain_point": cp, "order": list(o)} for cp, o in hits[:4]], "n_matches": len(hits)}
        if not hits:
            return None
        if len(hits) > 1:
            unique = False
        ident = tuple(range(identity_len))
        best = sorted(hits, key=lambda h: (h[1] != ident, h[0]))[0]
        return best

    # DLogProof: z G + c pk == R  <=>  c pk == R - z G
    dl = c["dlog"]["proof"]
    pk, Rr, z = W.point_from_json(dl["pk"]), W.point_from_json(dl["pk_t_rand_commitment"]), W.scalar_from_json(dl["challenge_response"])
    X = _sub(Rr, R.ec_mul(z, R.G))
    hit = settle("dlog", _search_points([Rr, R.G, pk], pk, X), 3)
    if hit:
        prof["chain_point"], prof["ord_dlog"] = hit[0], hit[1]
    # PedersenProof: z1 g + z2 h == a1 + a2 + e com
    pe = c["pedersen"]["proof"]
    com, a1, a2 = (W.point_from_json(pe[f]) for f in ("com", "a1", "a2"))
    z1, z2 = W.scalar_from_json(pe["z1"]), W.scalar_from_json(pe["z2"])
    X = _sub(_sub(R.ec_add(R.ec_mul(z1, R.G), R.ec_mul(z2, R.H2)), a1), a2)
    hit = settle("pedersen", _search_points([R.G, R.H2, com, a1, a2], com, X), 5)
    if hit:
        prof["ord_pedersen"] = hit[1]
        report["pedersen"]["chain_point_agrees_with_dlog"] = hit[0] == prof.get("chain_point", hit[0])
    # HomoELGamalProof: z2 G == A3 + e E  (and z1 H + z2 Y == T + e D, checked on the survivors)
    he, hp = c["heg"], c["heg"]["proof"]
    Gp, D, E = pt(he["G"]), pt(he["D"]), pt(he["E"])
    T, A3 = W.point_from_json(hp["T"]), W.point_from_json(hp["A3"])
    z1, z2 = W.scalar_from_json(hp["z1"]), W.scalar_from_json(hp["z2"])
    X2 = _sub(R.ec_mul(z2, Gp), A3)
    X1 = _sub(R.ec_add(R.ec_mul(z1, R.H2), R.ec_mul(z2, R.G)), T)
    hit = settle("heg", _search_points([T, A3, Gp, R.H2, R.G, D, E], E, X2, lambda e: R.ec_mul(e, D) == X1), 7)
    if hit:
        prof["ord_heg"] = hit[1]
    # ECDDHProof: z g1 == a1 + e h1, z g2 == a2 + e h2
    dd, dp = c["ecddh"], c["ecddh"]["proof"]
    g2, h1, h2 = pt(dd["g2"]), pt(dd["h1"]), pt(dd["h2"])
    a1, a2, z = W.point_from_json(dp["a1"]), W.point_from_json(dp["a2"]), W.scalar_from_json(dp["z"])
    X1, X2 = _sub(R.ec_mul(z, R.G), a1), _sub(R.ec_mul(z, g2), a2)
    hit = settle("ecddh", _search_points([R.G, h1, g2, h2, a1, a2], h1, X1, lambda e: R.ec_mul(e, h2) == X2), 6)
    if hit:
        prof["ord_ecddh"] = hit[1]
    # zk-paillier: NiCorrectKeyProof (zero encoding, mask order, salt), CompositeDLogProof (field order)
    k = c["keys"]
    N, Nt, h1n, h2n = H(k["N"]), H(k["Nt"]), H(k["h1"]), H(k["h2"])
    if "correct_key" in c:
        sig = [W.bigint_from_json(v) for v in c["correct_key"]["proof"]["sigma_vec"]]
        hits = []
        for zb in (0, 1):
            for mo in (0, 1):
                for salt in (0x4B5A656E, 0x6E655A4B, 0x5A656E4B):
                    with R.use_encoding(DEFAULT.replace(zero_bytes=zb, ck_mask_order=mo, ck_salt=salt)):
                        if all(pow(sig[i], N, N) == R.correct_key_rho(N, i) for i in (0, 1)):
                            hits.append((zb, mo, salt))
        report["correct_key"] = {"matches": [dict(zero_bytes=a, ck_mask_order=b, ck_salt=s) for a, b, s in hits], "n_matches": len(hits)}
        if hits:
            prof["zero_bytes"], prof["ck_mask_order"], prof["ck_salt"] = hits[0]
            unique = unique and len(hits) == 1
    if "composite_dlog" in c and c["composite_dlog"].get("verifies", True):
        cd = c["composite_dlog"]
        ni = H(cd["ni"]) if "ni" in cd else h2n
        x, y = W.bigint_from_json(cd["proof"]["x"]), W.bigint_from_json(cd["proof"]["y"])
        gy = pow(h1n, y, Nt)
        hits = []
        for order in itertools.permutations(range(4)):
            with R.use_encoding(DEFAULT.replace(ord_cdlog=order, zero_bytes=prof.get("zero_bytes", 0))):
                if gy * pow(ni, R.cdlog_digest(x, h1n, Nt, ni), Nt) % Nt == x:
                    hits.append(order)
        report["composite_dlog"] = {"matches": [list(o) for o in hits], "n_matches": len(hits)}
        if hits:
            prof["ord_cdlog"] = sorted(hits, key=lambda o: o != (0, 1, 2, 3))[0]
    missing = [name for name in ("dlog", "pedersen", "heg", "ecddh") if not report[name]["n_matches"]]
    missing += [name for name in ("correct_key", "composite_dlog") if name in report and not report[name]["n_matches"]]
    report["unique"] = unique
    report["no_combination_for"] = missing
    if missing:
        return None, report
    return R.Encoding(**prof), report
