#!/bin/bash
# One command on any machine WITH cargo (this image has none):
#     tools/rust_vectors/run.sh /path/to/multi-party-ecdsa        # a checkout of ZenGo-X/multi-party-ecdsa v0.8.1
# builds the reference's own test binary with dump_vectors.rs added, runs it, and leaves tests/golden/ref_vectors.json in
# THIS repository.  tests/test_ref_vectors_cpu.py (oracle) and tests/test_ref_vectors_gpu.py (HIP engine) then consume it:
# every proof the real crates generated must be accepted and every deterministic value reproduced byte for byte.
# Nothing is copied from the reference into this repository; the reference checkout gets one extra file under tests/.
set -euo pipefail
REF=${1:?usage: run.sh <reference checkout>}
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
command -v cargo >/dev/null || { echo "cargo not found: run this on a machine with a Rust toolchain"; exit 2; }
[ -f "$REF/Cargo.toml" ] && grep -q 'name = "multi-party-ecdsa"' "$REF/Cargo.toml" || { echo "$REF is not the reference crate"; exit 2; }
mkdir -p "$REF/tests"
cp "$HERE/dump_vectors.rs" "$REF/tests/dump_vectors.rs"
# dev-dependencies the dump needs (the reference already has serde_json and sha2; hex and zk-paillier / paillier are regular deps)
grep -q '^hex *=' "$REF/Cargo.toml" || sed -i 's/^\[dev-dependencies\]$/[dev-dependencies]\nhex = "0.4"/' "$REF/Cargo.toml"
grep -q '^serde_json *=' "$REF/Cargo.toml" || sed -i 's/^\[dev-dependencies\]$/[dev-dependencies]\nserde_json = "1"/' "$REF/Cargo.toml"
# The dump uses the crates' own OsRng sampling: vectors differ from run to run, which is fine — the consumers check
# "crate-generated proof is accepted" and "deterministic function of the dumped inputs is reproduced", never fixed bytes.
( cd "$REF" && cargo test --release --test dump_vectors -- --nocapture ) | grep '^{"schema"' > "$ROOT/tests/golden/ref_vectors.json"
python3 - "$ROOT/tests/golden/ref_vectors.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("ref_vectors.json:", d["crate"], "schema", d["schema"], len(d["cases"]), "cases")
PY
echo "now:  python -m pytest tests/test_ref_vectors_cpu.py -q      (and on a GPU box: -m gpu tests/test_ref_vectors_gpu.py)"
