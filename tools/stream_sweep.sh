#!/bin/bash
# On the GPU box: bench.py's c4_stream_1024 section (a stream of 1 024-session GG20 batches, several in flight on separate host threads /
# contexts / streams) against
#   QUEUES  the hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES; the runtime's default is 4, 20+ abort in the runtime)
#   DEPTHS  batches in flight
#   HINTS   nohint | hint  (mpe_ctx_set_device_share on every context)
#   BATCHES batches per run,  REPS repetitions
# e.g.  QUEUES="4 8 16" DEPTHS="2 3 8" tools/stream_sweep.sh      One line per run; full JSON lines under gpurun_out/stream/.
# (profiles/r04/stream_sweep*.log are runs of this script and its earlier forms; each log's first lines name the settings.)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/stream
echo "# QUEUES=${QUEUES:=16} DEPTHS=${DEPTHS:=2 3 8} HINTS=${HINTS:=nohint} BATCHES=${BATCHES:=48} REPS=${REPS:=1}"
for hwq in $QUEUES; do for hint in $HINTS; do for k in $DEPTHS; do for rep in $(seq 1 $REPS); do
  flag=""; [ $hint = hint ] && flag="--share-hint"
  name=q${hwq}_${hint}_k${k}_r$rep
  GPU_MAX_HW_QUEUES=$hwq timeout 300 python bench.py --stream-child --no-cpu-baseline --stream-batches $BATCHES \
      --stream-inflight $k $flag > gpurun_out/stream/$name.json 2> gpurun_out/stream/$name.err
  python3 - $name <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/stream/{n}.json") if l.startswith("{")][-1])
    c = d
    print(f"{n:22s} {c['signatures_per_s']:9.1f} sig/s  {c['ms_per_batch_sustained']:7.2f} ms/batch  signed {c['all_sessions_signed']} ossl {c['openssl_verified']}/{c['openssl_of']}")
except Exception as e:
    print(n, "FAILED", repr(e), open(f"gpurun_out/stream/{n}.err").read()[-200:].replace("\n", " "))
PY
done; done; done; done
