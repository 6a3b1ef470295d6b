"""Generates tests/golden/gg20_sessions.json: every round message and the signature of complete GG20 signing sessions,
computed by the pure-Python restatement tests/pyref_gg20.py ALONE (no oracle, no GPU).  Inputs come from the seeded
fixtures (tests/gg20_fixture.py), so the file only stores outputs.  Run from the repo root:
    python tests/golden/make_golden_gg20.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fixtures as F  # noqa: E402
import gg20_fixture as G  # noqa: E402

CASES = {"t1n3": dict(t=1, n=3, signers=[0, 1], B=2, seed="golden-t1n3"),
         "t2n5": dict(t=2, n=5, signers=[0, 2, 4], B=1, seed="golden-t2n5")}


def main():
    keys = F.load_keys()
    out = {}
    for name, c in CASES.items():
        lk = G.make_local_keys(keys, c["t"], c["n"], c["signers"])
        nonces = G.make_nonces(lk, c["B"], seed=c["seed"])
        msgs, sigs = [], []
        for b in range(c["B"]):
            packed, sg, pst = G.py_session(lk, nonces, b)
            assert all(st == (0, []) for st in pst) and all(x == sg[0] for x in sg)
            msgs.append({str(rnd): [m.hex() for m in packed[rnd]] for rnd in G.ROUNDS})
            sigs.append([hex(sg[0][0]), hex(sg[0][1]), sg[0][2]])
        out[name] = dict(c, messages=msgs, sig=sigs)
    with open(os.path.join(HERE, "gg20_sessions.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
