#!/bin/bash
# On the GPU box: the headline step with every library variant under tools/ab/, back to back on the same device
# (bench.py --no-configs --no-cpu-baseline).  Prints one line per variant; full JSON lines under gpurun_out/ab/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
for rep in 1 2; do
for so in tools/ab/*.so; do
  name=$(basename $so .so)
  unset MPE_NO_SLIDING; case $name in *noslide*) export MPE_NO_SLIDING=1;; esac
  MPE_LIB_PATH=$PWD/$so python bench.py --steps ${STEPS:-3} --warmup 1 --no-configs --no-cpu-baseline ${AB_ARGS} > gpurun_out/ab/$name.$rep.json 2> gpurun_out/ab/$name.$rep.err
  python3 - "$name" "$rep" <<'PY'
import json, sys
name, rep = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f"gpurun_out/ab/{name}.{rep}.json") if l.startswith("{")][-1])
    sec = {s["kernel"].split("<")[-1][:18] + ("h" if "half" in s["kernel"] else ""): round(s["seconds"] / d["steps"], 4) for s in d.get("roofline_secondary", [])}
    print(f"{name:14s} rep {rep}: {d['value']:9.1f} sig/s  step {d['ms_per_step']:8.1f} ms  dom avg {d['roofline']['avg_kernel_ms']:8.2f} ms  frac {d['roofline']['frac']:.4f}  signed {d['all_sessions_signed']} ossl {d.get('openssl_verified')}  {sec}")
except Exception as e:
    print(name, rep, "FAILED", e)
PY
done
done
