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
# Pin the three un-vendored crates to the exact releases the reference's manifest names (Cargo.toml:36-47 gives "0.9", "0.4.3",
# "0.4.2"; the reference ignores its Cargo.lock, .gitignore:7).  The lock file cargo resolves is copied next to the vectors, so
# the pin travels with them: tests/golden/ref_vectors.Cargo.lock is the record of what produced ref_vectors.json.
( cd "$REF" && cargo generate-lockfile \
   && { cargo update -p curv-kzen --precise "${CURV_VERSION:-0.9.0}" || true; } \
   && { cargo update -p kzen-paillier --precise "${PAILLIER_VERSION:-0.4.2}" || true; } \
   && { cargo update -p zk-paillier --precise "${ZKP_VERSION:-0.4.3}" || true; } )
# The dump uses the crates' own OsRng sampling: vectors differ from run to run, which is fine — the consumers check
# "crate-generated proof is accepted" and "deterministic function of the dumped inputs is reproduced", never fixed bytes.
( cd "$REF" && cargo test --release --test dump_vectors -- --nocapture ) | grep '^{"schema"' > "$ROOT/tests/golden/ref_vectors.json"
cp "$REF/Cargo.lock" "$ROOT/tests/golden/ref_vectors.Cargo.lock"
python3 - "$ROOT/tests/golden/ref_vectors.json" "$REF/Cargo.lock" <<'PY'
import json, re, sys
d = json.load(open(sys.argv[1]))
lock = open(sys.argv[2]).read()
vers = {}
for name in ("curv-kzen", "kzen-paillier", "zk-paillier", "rust-gmp-kzen", "secp256k1", "sha2", "multi-party-ecdsa"):
    m = re.search(r'name = "%s"\nversion = "([^"]+)"' % re.escape(name), lock)
    vers[name] = m.group(1) if m else None
d["versions"] = vers
json.dump(d, open(sys.argv[1], "w"))
print("ref_vectors.json:", d["crate"], "schema", d["schema"], len(d["cases"]), "cases; crates:", vers)
PY
# which byte-level conventions do these crates use?  (prints the profile; writes tests/golden/encoding_profile.json)
python3 "$ROOT/tools/diagnose_encodings.py" "$ROOT/tests/golden/ref_vectors.json"
echo "now:  python -m pytest tests/test_ref_vectors_cpu.py -q      (and on a GPU box: -m gpu tests/test_ref_vectors_gpu.py)"
