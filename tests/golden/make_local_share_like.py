"""Generates tests/golden/local_share_like.json: party 2's LocalKey of the t=1, n=3 fixture wallet in the layout
`gg20_keygen` writes (examples/gg20_keygen.rs:52-56; LocalKey, state_machine/keygen/rounds.rs:311-322), in the serde forms
curv 0.9 / kzen-paillier 0.4.2 are BELIEVED to use (wire.DEFAULT_STYLE: byte arrays for Point / Scalar, decimal strings for
the Paillier key, hex strings for curv BigInt).  Produced by this repository's own encoder, NOT by the Rust crates: it pins
the decoder's handling of that layout, not the crates' bytes (tools/rust_vectors is what would).
Run from the repo root:  python tests/golden/make_local_share_like.py"""
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
import fixtures as F          # noqa: E402
import gg20_fixture as G      # noqa: E402

keys = F.load_keys()
lk = G.make_local_keys(keys, 1, 3, [0, 1])
A = lk["arrays"]
xs, X, y = F.ints(A["x"]), F.points(A["X"]), F.points(A["y"])[0]
Ns, stm = [k.N for k in lk["keys"]], [(k.Nt, k.h1, k.h2) for k in lk["keys"]]
doc = W.local_key_to_json(2, 1, 3, lk["keys"][1].p, lk["keys"][1].q, xs[1], y, X, Ns, stm, vss_commitments=[y, X[0]])
with open(os.path.join(HERE, "local_share_like.json"), "w") as f:
    json.dump(doc, f, indent=1)
