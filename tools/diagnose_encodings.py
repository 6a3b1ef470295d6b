#!/usr/bin/env python3
"""Which byte-level conventions do the REAL crates use?

    tools/rust_vectors/run.sh /path/to/multi-party-ecdsa      # on a machine with cargo: writes tests/golden/ref_vectors.json
    python tools/diagnose_encodings.py [vectors.json]         # here: prints the profile, writes tests/golden/encoding_profile.json

For every proof in the file the script tries EVERY combination this repository knows — both forms of DigestExt::chain_point,
every order of the points inside the challenge of DLogProof / PedersenProof / HomoELGamalProof / ECDDHProof, both encodings of
BigInt zero, both block orders of zk-paillier's mask generation, three byte orders of its salt, every field order of
CompositeDLogProof — with hashlib and Python integers only, and reports the one under which the crate-generated proof verifies.
Expected outcome: "profile == defaults: True" (the defaults of include/mpecdsa_hip.h are what the crates do: nothing to change).
Otherwise the printed profile is what a host installs with `mpe_ctx_set_encoding` (Python harness: `Context(0, encoding=...)`,
`enc_profiles.load_profile`); no kernel is edited, no library rebuilt.  Exit status 1 only when some proof verifies under NO
combination (then a recalled formula, not an encoding, is wrong — the report names the proof)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import enc_profiles as ENCS   # noqa: E402

spec = importlib.util.spec_from_file_location("mpe_wire", os.path.join(ROOT, "multi_party_ecdsa_amd", "wire.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ref_vectors.json")
    if not os.path.exists(path):
        print(f"{path} does not exist: produce it with tools/rust_vectors/run.sh on a machine with cargo")
        return 2
    doc = json.load(open(path))
    print(f"{os.path.relpath(path, ROOT)}: {doc.get('crate')}  schema {doc.get('schema')}  {len(doc['cases'])} cases")
    prof, report = ENCS.diagnose(doc["cases"], wire=W)
    for name in ("dlog", "pedersen", "heg", "ecddh", "correct_key", "composite_dlog"):
        if name in report:
            print(f"  {name:15s} combinations that verify: {report[name]['n_matches']:3d}   {report[name]['matches'][:2]}")
    if prof is None:
        print("NO combination verifies:", report["no_combination_for"], "- a recalled formula differs, not an encoding")
        return 1
    # the same profile must hold for every case of the file
    for i, c in enumerate(doc["cases"][1:], 1):
        p_i, _ = ENCS.diagnose([c], wire=W)
        if p_i != prof:
            print(f"case {i} gives a different profile: {p_i!r}")
            return 1
    print("profile:", repr(prof))
    print("profile == defaults:", prof == ENCS.DEFAULT, " unique:", report["unique"])
    out = os.path.join(ROOT, "tests", "golden", "encoding_profile.json")
    if "SELF-MADE" not in str(doc.get("crate")):
        with open(out, "w") as f:
            json.dump({"source": os.path.basename(path), "crate": doc.get("crate"), "profile": prof.as_dict()}, f, indent=1)
        print("written:", os.path.relpath(out, ROOT))
    return 0


if __name__ == "__main__":
    sys.exit(main())
