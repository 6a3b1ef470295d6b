#!/bin/bash
# A/B of the persistent-grid rule (mpe_internal.h persistent_grid), same box: equal trips (rounds 1-4) | full trips + a tail of lone
# waves | hybrid (the tail only when it fits one wave per SIMD: the default) on config 5's per-GPU share, mid-size t=1 n=3 batches and the
# pipelined engine; the launches of the timed region (kind, bits, batch, ms) ride along so that the launch that moves can be named.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/abgrid
one() {  # label, mode, bench args
  local label=$1 mode=$2; shift 2
  MPE_GRID=$mode python bench.py --no-cpu-baseline --no-configs --warmup 1 --dump-launches "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
heavy=[x for x in d['launches_timed_region'] if x['kind'] in (0,3,6) and x['ms']>2.0]
n=len(heavy)//d['steps']
print(json.dumps({'case':'$label','grid':'$mode','signatures_per_s':round(d['value'],1),'ms_per_step':round(d['ms_per_step'],2),'dominant_frac':round(d['roofline']['frac'],4),'whole_step_frac':round(d['whole_step']['frac'],4),
 'launches_last_step':[[x['kind'],x['bits'],x['exp_words'],x['batch'],x['ms']] for x in heavy[-n:]]}))" | tee -a gpurun_out/abgrid/ab.jsonl
}
for m in equal full hybrid; do
  one c5_t2n5_8192 $m --t 2 --n 5 --sessions 8192 --steps 2
  one t1n3_4096 $m --sessions 4096 --steps 3
  one t1n3_12288 $m --sessions 12288 --steps 2
  MPE_GRID=$m python tools/exp_pipeline.py --lanes 2 --group 4 --batches 96 2>/dev/null | tail -1 | sed "s/^{/{\"grid\": \"$m\", /" | tee -a gpurun_out/abgrid/ab.jsonl
done
