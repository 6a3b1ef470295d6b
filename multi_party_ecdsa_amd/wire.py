"""Wire form of the GG20 signing messages: the JSON `serde` gives the reference's message types, to and from the fixed-size
records of the round engine (include/mpecdsa_hip.h "GG20 round messages") — SURVEY.md §8f-1.  Host-side glue only (no
arithmetic): a relay process uses it to let GPU-held parties talk to Rust parties through `gg20_sm_manager`
(examples/gg20_sm_manager.rs:17-51, examples/gg20_sm_client.rs:10-53).

Field names and nesting come from the reference sources (all `#[derive(Serialize, Deserialize)]`):
  Msg{sender, receiver, body}                                   round-based 0.1.4 (examples/gg20_sm_client.rs:35-40)
  OfflineProtocolMessage(OfflineM), OfflineM::M1..M6            state_machine/sign.rs:478-490
  MessageA{c, range_proofs}, MessageB{c, b_proof, beta_tag_proof}   src/utilities/mta/mod.rs:34-45
  AliceProof{z, e, s, s1, s2}                                   src/utilities/mta/range_proofs.rs:94-101
  PDLwSlackProof{z, u1, u2, u3, s1, s2, s3}                     src/utilities/zk_pdl_with_slack/mod.rs:56-65
  SignBroadcastPhase1{com}, SignDecommitPhase1{blind_factor, g_gamma_i}, SignatureRecid{r, s, recid}   gg_2020/party_i.rs:111-135
  GammaI / WI / DeltaI / TI / TIProof / RDash / SI / HEGProof / PartialSignature: newtype structs    sign/rounds.rs:31-49,661
  DLogProof{pk, pk_t_rand_commitment, challenge_response}, PedersenProof{e, a1, a2, com, z1, z2},
  HomoELGamalProof{T, A3, z1, z2}                               curv-kzen 0.9 (un-vendored; field names recalled)

THE PRIMITIVE ENCODINGS ARE RECALLED, NOT READ (curv-kzen 0.9, kzen-paillier 0.4.2 are not in the reference tree), so the
ENCODERS are configurable (`Style`) and the DECODERS accept every form the crates are believed to have used:
  * curv `BigInt`: lower-case hex string of the big-endian bytes (human-readable serializers) — or a byte array;
  * curv `Point<Secp256k1>` / `Scalar<Secp256k1>`: {"curve": "secp256k1", "point" / "scalar": BYTES}, where serde_json writes
    bytes as an array of numbers (Style point="bytes", the default: what a real local-share.json is believed to hold) — or a
    hex string (point="hex"); points compressed (33 bytes), uncompressed accepted;
  * kzen-paillier `EncryptionKey{n}` / `DecryptionKey{p,q}`: its own `serialize::bigint` module, believed to write RADIX-10
    strings (Style paillier="decimal", the default) — or hex (paillier="hex").  A digit-only string is ambiguous between the
    two radices: `local_key_from_json` decodes both ways and keeps the reading in which p * q == paillier_key_vec[i-1].n.
Unknown fields are ignored.  They are isolated in the functions below, so that vectors produced by the real crates
(tools/rust_vectors) settle the form in one place."""
import json
from dataclasses import dataclass

import numpy as np

P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
CURVE = "secp256k1"


@dataclass(frozen=True)
class Style:
    """which of the believed serde forms the encoders write"""
    point: str = "bytes"        # "bytes": [2, 121, ...] (serde_json's rendering of serde bytes) | "hex": "0279be..."
    scalar: str = "bytes"       # likewise for Scalar
    bigint: str = "hex"         # curv BigInt: "hex" string | "bytes" array
    paillier: str = "decimal"   # kzen-paillier keys: "decimal" string | "hex" string

    def __post_init__(self):
        for f, allowed in (("point", ("bytes", "hex")), ("scalar", ("bytes", "hex")), ("bigint", ("hex", "bytes")), ("paillier", ("decimal", "hex"))):
            if getattr(self, f) not in allowed:
                raise ValueError(f"Style.{f} must be one of {allowed}")


DEFAULT_STYLE = Style()
HEX_STYLE = Style(point="hex", scalar="hex", bigint="hex", paillier="hex")     # the round-2 form of this module


# ---- curv / kzen-paillier primitives (recalled encodings) --------------------------------------------------------------
def bigint_to_json(x, style=DEFAULT_STYLE):
    h = "%x" % int(x)
    h = h if len(h) % 2 == 0 else "0" + h
    return list(bytes.fromhex(h)) if style.bigint == "bytes" else h


def bigint_from_json(v, radix=16, strict=False):
    """curv BigInt: hex string (radix=16) / byte array / int.  radix=10 reads a digit string as decimal (kzen-paillier).
    A digit-only string is BOTH a decimal and a hexadecimal numeral; curv 0.9 is believed to write hex (even length), which
    is what radix=16 reads.  strict=True refuses such a string when the two readings differ — for callers that have no
    cross-check (the Paillier key fields have one: p q = n, `paillier_bigint_readings`) and would rather fail than guess."""
    if isinstance(v, bool):
        raise ValueError("not a big integer")
    if isinstance(v, int):
        return v
    if isinstance(v, list):
        return int.from_bytes(bytes(v), "big")
    s = v.strip()
    if s.startswith("0x"):
        return int(s, 16)
    if strict and s.isdigit() and int(s, 10) != int(s, 16):
        raise ValueError(f"ambiguous big integer {s[:20]!r}: digit-only, decimal and hexadecimal readings differ")
    return int(s, radix)


def paillier_bigint_to_json(x, style=DEFAULT_STYLE):
    return str(int(x)) if style.paillier == "decimal" else "%x" % int(x)


def paillier_bigint_readings(v):
    """every integer a kzen-paillier key field may mean: [decimal reading, hex reading] for a digit-only string"""
    if isinstance(v, (int, list)) and not isinstance(v, bool):
        return [bigint_from_json(v)]
    s = v.strip()
    out = []
    if s.isdigit():
        out.append(int(s, 10))
    try:
        h = int(s, 16)
        if h not in out:
            out.append(h)
    except ValueError:
        pass
    if not out:
        raise ValueError("not a big integer")
    return out


def _bytes_json(b, how):
    return list(b) if how == "bytes" else b.hex()


def point_to_json(pt, style=DEFAULT_STYLE):
    x, y = pt
    return {"curve": CURVE, "point": _bytes_json(bytes([2 + (y & 1)]) + int(x).to_bytes(32, "big"), style.point)}


def point_from_json(v):
    raw = v["point"] if isinstance(v, dict) else v
    if isinstance(v, dict) and v.get("curve", CURVE) != CURVE:
        raise ValueError("point of another curve")
    b = bytes(raw) if isinstance(raw, list) else bytes.fromhex(raw)
    if len(b) == 65 and b[0] == 4:
        x, y = int.from_bytes(b[1:33], "big"), int.from_bytes(b[33:], "big")
        if (y * y - x * x * x - 7) % P or x >= P or y >= P:
            raise ValueError("point is not on the curve")
        return x, y
    if len(b) == 33 and b[0] in (2, 3):
        x = int.from_bytes(b[1:], "big")
        y = pow((x * x * x + 7) % P, (P + 1) // 4, P)
        if (y * y - x * x * x - 7) % P or x >= P:
            raise ValueError("point is not on the curve")
        return (x, y if (y & 1) == (b[0] & 1) else P - y)
    raise ValueError("unknown point encoding")


def scalar_to_json(x, style=DEFAULT_STYLE):
    return {"curve": CURVE, "scalar": _bytes_json(int(x).to_bytes(32, "big"), style.scalar)}


def scalar_from_json(v):
    raw = v["scalar"] if isinstance(v, dict) else v
    return int.from_bytes(bytes(raw), "big") if isinstance(raw, list) else int(raw, 16)


# ---- words <-> ints ----------------------------------------------------------------------------------------------------
def _int(words):
    return int.from_bytes(np.ascontiguousarray(words, dtype="<u4").tobytes(), "little")


def _pt(words):
    v = _int(words)
    return v & ((1 << 256) - 1), v >> 256


def _put(rec, off, n, value):
    rec[off:off + n] = np.frombuffer(int(value).to_bytes(4 * n, "little"), dtype="<u4")


def _put_pt(rec, off, pt):
    _put(rec, off, 8, pt[0])
    _put(rec, off + 8, 8, pt[1])


ALICE = (("z", 0, 64), ("e", 64, 8), ("s", 72, 64), ("s1", 136, 25), ("s2", 161, 89))
PDL = (("z", 0, 64), ("u2", 80, 128), ("u3", 208, 64), ("s1", 272, 25), ("s2", 297, 64), ("s3", 361, 89))


def _dlog_to_json(w, off, style=DEFAULT_STYLE):
    return {"pk": point_to_json(_pt(w[off:off + 16]), style), "pk_t_rand_commitment": point_to_json(_pt(w[off + 16:off + 32]), style),
            "challenge_response": scalar_to_json(_int(w[off + 32:off + 40]), style)}


def _dlog_from_json(rec, off, j):
    _put_pt(rec, off, point_from_json(j["pk"]))
    _put_pt(rec, off + 16, point_from_json(j["pk_t_rand_commitment"]))
    _put(rec, off + 32, 8, scalar_from_json(j["challenge_response"]))


def _msgb_to_json(w, style=DEFAULT_STYLE):
    return {"c": bigint_to_json(_int(w[0:128]), style), "b_proof": _dlog_to_json(w, 128, style), "beta_tag_proof": _dlog_to_json(w, 168, style)}


def _msgb_from_json(rec, off, j):
    _put(rec, off, 128, bigint_from_json(j["c"]))
    _dlog_from_json(rec, off + 128, j["b_proof"])
    _dlog_from_json(rec, off + 168, j["beta_tag_proof"])


def record_to_bodies(rnd, rec, S, n, sender, style=DEFAULT_STYLE):
    """One sender's record of round `rnd` (0..5, 7) -> list of (receiver or None, body) where body is the JSON value of
    `OfflineProtocolMessage` (rounds 0..5) / `PartialSignature` (round 7).  sender, receiver: signer ordinals + 1 (the
    reference numbers parties from 1)."""
    w = np.ascontiguousarray(rec, dtype=np.uint32)
    if rnd == 0:
        proofs = [{f: bigint_to_json(_int(w[st * 256 + o:st * 256 + o + k]), style) for f, o, k in ALICE} for st in range(n)]
        c = w[n * 256:]
        return [(None, {"M1": [{"c": bigint_to_json(_int(c[0:128]), style), "range_proofs": proofs}, {"com": bigint_to_json(_int(c[128:136]), style)}]})]
    if rnd == 1:
        out = []
        for jj in range(S - 1):
            ind = jj if jj < sender - 1 else jj + 1
            g, wi = w[(jj * 2) * 208:(jj * 2 + 1) * 208], w[(jj * 2 + 1) * 208:(jj * 2 + 2) * 208]
            out.append((ind + 1, {"M2": [_msgb_to_json(g, style), _msgb_to_json(wi, style)]}))
        return out
    if rnd == 2:
        proof = {"e": scalar_to_json(_int(w[24:32]), style), "a1": point_to_json(_pt(w[32:48]), style), "a2": point_to_json(_pt(w[48:64]), style),
                 "com": point_to_json(_pt(w[64:80]), style), "z1": scalar_to_json(_int(w[80:88]), style), "z2": scalar_to_json(_int(w[88:96]), style)}
        return [(None, {"M3": [scalar_to_json(_int(w[0:8]), style), point_to_json(_pt(w[8:24]), style), proof]})]
    if rnd == 3:
        return [(None, {"M4": {"blind_factor": bigint_to_json(_int(w[0:8]), style), "g_gamma_i": point_to_json(_pt(w[8:24]), style)}})]
    if rnd == 4:
        proofs = []
        for jj in range(S - 1):
            p = w[jj * 450:(jj + 1) * 450]
            d = {f: bigint_to_json(_int(p[o:o + k]), style) for f, o, k in PDL}
            d["u1"] = point_to_json(_pt(p[64:80]), style)
            proofs.append({f: d[f] for f in ("z", "u1", "u2", "u3", "s1", "s2", "s3")})
        return [(None, {"M5": [point_to_json(_pt(w[(S - 1) * 450:(S - 1) * 450 + 16]), style), proofs]})]
    if rnd == 5:
        proof = {"T": point_to_json(_pt(w[16:32]), style), "A3": point_to_json(_pt(w[32:48]), style), "z1": scalar_to_json(_int(w[48:56]), style),
                 "z2": scalar_to_json(_int(w[56:64]), style)}
        return [(None, {"M6": [point_to_json(_pt(w[0:16]), style), proof]})]
    if rnd == 7:
        return [(None, scalar_to_json(_int(w[0:8]), style))]
    raise ValueError(rnd)


def record_to_msgs(rnd, rec, S, n, sender, style=DEFAULT_STYLE):
    """`Msg<OfflineProtocolMessage>` values as the relay carries them (JSON strings)"""
    return [json.dumps({"sender": sender, "receiver": r, "body": body}) for r, body in record_to_bodies(rnd, rec, S, n, sender, style)]


def bodies_to_record(rnd, bodies, S, n, sender):
    """inverse of record_to_bodies: the messages one sender emitted in round `rnd` -> its record (uint32 words)"""
    W = {0: 256 * (n + 1), 1: 208 * 2 * (S - 1), 2: 96, 3: 24, 4: 450 * S, 5: 64, 7: 8}[rnd]
    rec = np.zeros(W, dtype=np.uint32)
    if rnd == 0:
        ma, bc = bodies[0][1]["M1"]
        for st, pr in enumerate(ma["range_proofs"]):
            for f, o, k in ALICE:
                _put(rec, st * 256 + o, k, bigint_from_json(pr[f]))
        _put(rec, n * 256, 128, bigint_from_json(ma["c"]))
        _put(rec, n * 256 + 128, 8, bigint_from_json(bc["com"]))
    elif rnd == 1:
        for receiver, body in bodies:
            ind = receiver - 1
            jj = ind if ind < sender - 1 else ind - 1
            g, wi = body["M2"]
            _msgb_from_json(rec, (jj * 2) * 208, g)
            _msgb_from_json(rec, (jj * 2 + 1) * 208, wi)
    elif rnd == 2:
        delta, T, pr = bodies[0][1]["M3"]
        _put(rec, 0, 8, scalar_from_json(delta)); _put_pt(rec, 8, point_from_json(T))
        _put(rec, 24, 8, scalar_from_json(pr["e"])); _put_pt(rec, 32, point_from_json(pr["a1"])); _put_pt(rec, 48, point_from_json(pr["a2"]))
        _put_pt(rec, 64, point_from_json(pr["com"])); _put(rec, 80, 8, scalar_from_json(pr["z1"])); _put(rec, 88, 8, scalar_from_json(pr["z2"]))
    elif rnd == 3:
        d = bodies[0][1]["M4"]
        _put(rec, 0, 8, bigint_from_json(d["blind_factor"])); _put_pt(rec, 8, point_from_json(d["g_gamma_i"]))
    elif rnd == 4:
        rdash, proofs = bodies[0][1]["M5"]
        for jj, pr in enumerate(proofs):
            for f, o, k in PDL:
                _put(rec, jj * 450 + o, k, bigint_from_json(pr[f]))
            _put_pt(rec, jj * 450 + 64, point_from_json(pr["u1"]))
        _put_pt(rec, (S - 1) * 450, point_from_json(rdash))
    elif rnd == 5:
        Si, pr = bodies[0][1]["M6"]
        _put_pt(rec, 0, point_from_json(Si)); _put_pt(rec, 16, point_from_json(pr["T"])); _put_pt(rec, 32, point_from_json(pr["A3"]))
        _put(rec, 48, 8, scalar_from_json(pr["z1"])); _put(rec, 56, 8, scalar_from_json(pr["z2"]))
    elif rnd == 7:
        _put(rec, 0, 8, scalar_from_json(bodies[0][1]))
    else:
        raise ValueError(rnd)
    return rec


def signature_to_json(r, s, recid, style=DEFAULT_STYLE):
    """`SignatureRecid` (party_i.rs:131-135)"""
    return {"r": scalar_to_json(r, style), "s": scalar_to_json(s, style), "recid": int(recid)}


# ---- LocalKey (state_machine/keygen/rounds.rs:311-322) -------------------------------------------------------------------
# What `gg20_keygen` writes to local-share{i}.json (examples/gg20_keygen.rs:52-56) and `gg20_signing` reads back: the serde
# form of LocalKey<Secp256k1>{paillier_dk{p,q}, pk_vec, keys_linear{y,x_i}, paillier_key_vec[{n}], y_sum_s,
# h1_h2_n_tilde_vec[{N,g,ni}], vss_scheme{parameters{threshold,share_count},commitments}, i, t, n}.  `i` is 1-based.
def local_key_to_json(i, t, n, p, q, x_i, y, pk_vec, N_vec, stm_vec, vss_commitments=None, style=DEFAULT_STYLE):
    """i: 1-based party index; stm_vec: [(N~, h1, h2)] per party; vss_commitments: t+1 points (optional: signing never reads them)"""
    return {
        "paillier_dk": {"p": paillier_bigint_to_json(p, style), "q": paillier_bigint_to_json(q, style)},
        "pk_vec": [point_to_json(P_, style) for P_ in pk_vec],
        "keys_linear": {"y": point_to_json(y, style), "x_i": scalar_to_json(x_i, style)},
        "paillier_key_vec": [{"n": paillier_bigint_to_json(N_, style)} for N_ in N_vec],
        "y_sum_s": point_to_json(y, style),
        "h1_h2_n_tilde_vec": [{"N": bigint_to_json(a, style), "g": bigint_to_json(b, style), "ni": bigint_to_json(c, style)} for a, b, c in stm_vec],
        "vss_scheme": {"parameters": {"threshold": t, "share_count": n}, "commitments": [point_to_json(c, style) for c in (vss_commitments or [])]},
        "i": i, "t": t, "n": n,
    }


def local_key_from_json(obj):
    """-> dict(i (1-based), t, n, p, q, x_i, y, pk_vec, N_vec, stm_vec) of Python ints / (x, y) points.
    The kzen-paillier fields (paillier_dk.p/q, paillier_key_vec[].n) are read in the radix that makes p * q == n[i-1] and
    every n a 2047/2048-bit integer (party_i.rs:287-290 enforces that size on receipt): decimal first, then hex."""
    if isinstance(obj, (str, bytes)):
        obj = json.loads(obj)
    i, n = int(obj["i"]), int(obj["n"])
    ek = lambda e: e["n"] if isinstance(e, dict) else e
    pr, qr = paillier_bigint_readings(obj["paillier_dk"]["p"]), paillier_bigint_readings(obj["paillier_dk"]["q"])
    if not (len(obj["paillier_key_vec"]) == n and 1 <= i <= n):
        raise ValueError("LocalKey: inconsistent vector lengths / indices")
    nr = [paillier_bigint_readings(ek(e)) for e in obj["paillier_key_vec"]]
    pick = None
    for p_ in pr:
        for q_ in qr:
            if p_ * q_ in nr[i - 1]:
                pick = (p_, q_)
                break
        if pick:
            break
    if pick is None:
        raise ValueError("LocalKey: paillier_dk does not match paillier_key_vec[i-1] in any radix")
    # the other parties' moduli: the reading with the size the reference accepts; both radices plausible -> the one my own key used
    mine_decimal = isinstance(obj["paillier_dk"]["p"], str) and obj["paillier_dk"]["p"].strip().isdigit() and pick[0] == pr[0]
    N_vec = []
    for k, cand in enumerate(nr):
        good = [c for c in cand if c.bit_length() in (2047, 2048)] or cand
        if k == i - 1:
            good = [pick[0] * pick[1]]
        N_vec.append(good[0] if (mine_decimal or len(good) == 1) else good[-1])
    out = dict(i=i, t=int(obj["t"]), n=n, p=pick[0], q=pick[1], x_i=scalar_from_json(obj["keys_linear"]["x_i"]),
               y=point_from_json(obj["y_sum_s"]), pk_vec=[point_from_json(v) for v in obj["pk_vec"]], N_vec=N_vec,
               stm_vec=[(bigint_from_json(s_["N"]), bigint_from_json(s_["g"]), bigint_from_json(s_["ni"])) for s_ in obj["h1_h2_n_tilde_vec"]])
    if not (len(out["pk_vec"]) == len(out["stm_vec"]) == n and out["t"] < n):
        raise ValueError("LocalKey: inconsistent vector lengths / indices")
    return out


def _words(vals, nwords):
    out = np.zeros((len(vals), nwords), dtype=np.uint32)
    for r, v in enumerate(vals):
        _put(out[r], 0, nwords, v)
    return out


def _pt_words(pts):
    out = np.zeros((len(pts), 16), dtype=np.uint32)
    for r, pt in enumerate(pts):
        _put_pt(out[r], 0, pt)
    return out


def local_keys_to_arrays(local_keys):
    """One or more LocalKeys of ONE wallet (one per party this process acts for; dicts from local_key_from_json) -> the arrays of
    mpe_gg20_keys_create / engine.Gg20Keys plus `own` (0-based party indices whose secrets are present).  Rows of x, p, q of the
    parties that are not present stay zero: build the key object with own=... and they never reach the device."""
    lk0 = local_keys[0]
    n, t = lk0["n"], lk0["t"]
    for lk in local_keys[1:]:
        same = all(lk[f] == lk0[f] for f in ("t", "n", "y", "pk_vec", "N_vec", "stm_vec"))
        if not same:
            raise ValueError("LocalKeys of different wallets")
    xs, ps, qs = [0] * n, [0] * n, [0] * n
    for lk in local_keys:
        xs[lk["i"] - 1], ps[lk["i"] - 1], qs[lk["i"] - 1] = lk["x_i"], lk["p"], lk["q"]
    arrays = dict(x=_words(xs, 8), p=_words(ps, 32), q=_words(qs, 32), N=_words(lk0["N_vec"], 64),
                  Nt=_words([s[0] for s in lk0["stm_vec"]], 64), h1=_words([s[1] for s in lk0["stm_vec"]], 64),
                  h2=_words([s[2] for s in lk0["stm_vec"]], 64), y=_pt_words([lk0["y"]]), X=_pt_words(lk0["pk_vec"]))
    return dict(t=t, n=n, arrays=arrays, own=sorted(lk["i"] - 1 for lk in local_keys))
