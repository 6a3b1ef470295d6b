#!/bin/bash
# On the GPU box: bench.py's c4_stream_1024 section (a stream of 1 024-session batches, several in flight on separate host threads /
# contexts / streams) against (a) the number of batches in flight, (b) the device-share hint of the contexts (mpe_ctx_set_device_share:
# keep the efficient lane layouts because other batches fill the idle lanes), (c) the number of hardware queues the HIP runtime uses.
# One line per run; full JSON under gpurun_out/stream/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/stream
for hwq in 4 8; do
for hint in hint nohint; do
for k in 2 3 4; do
  flag=""; [ $hint = hint ] && flag="--share-hint"
  name=q${hwq}_${hint}_k$k
  GPU_MAX_HW_QUEUES=$hwq timeout 200 python bench.py --sessions 1024 --steps 1 --warmup 0 --no-cpu-baseline --only c4_stream --stream-batches ${BATCHES:-24} \
      --stream-inflight $k $flag > gpurun_out/stream/$name.json 2> gpurun_out/stream/$name.err
  python3 - $name <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/stream/{n}.json") if l.startswith("{")][-1])
    c = d["configs"]["c4_stream_1024"]
    print(f"{n:18s} {c['signatures_per_s']:9.1f} sig/s  {c['ms_per_batch_sustained']:7.2f} ms/batch  signed {c['all_sessions_signed']} ossl {c['openssl_verified']}/{c['openssl_of']}   (single 1024 batch, same process: {d['value']:.0f} sig/s)")
except Exception as e:
    print(n, "FAILED", e)
PY
done; done; done
