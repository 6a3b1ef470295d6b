#!/bin/bash
# A/B of the persistent-grid rule (mpe_internal.h persistent_grid): full trips + a tail of lone waves (default) against equal trips
# (MPE_GRID_EQUAL=1, rounds 1-4), same box: config 5's per-GPU share, the pipelined engine, one 4 096-session pass, the headline.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/abgrid
one() {  # label, env, bench args
  local label=$1 envs=$2; shift 2
  env $envs python bench.py --no-cpu-baseline --no-configs --warmup 1 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'case':'$label','env':'$envs','signatures_per_s':round(d['value'],1),'ms_per_step':round(d['ms_per_step'],2),'dominant_frac':round(d['roofline']['frac'],4),'whole_step_frac':round(d['whole_step']['frac'],4),
 'secondary':{x['kernel'][:40]:round(x['frac'],3) for x in d['roofline_secondary']}}))" | tee -a gpurun_out/abgrid/ab.jsonl
}
for e in "MPE_X=0" "MPE_GRID_EQUAL=1"; do
  one c5_t2n5_8192 "$e" --t 2 --n 5 --sessions 8192 --steps 2
  one t1n3_4096 "$e" --sessions 4096 --steps 3
  one t1n3_12288 "$e" --sessions 12288 --steps 2
  one headline_65536 "$e" --steps 1
  env $e python tools/exp_pipeline.py --lanes 2 --group 4 --batches 96 2>/dev/null | tail -1 | tee -a gpurun_out/abgrid/ab.jsonl
done
