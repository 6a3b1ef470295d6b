#!/bin/bash
# Same-box A/B of (1) the threshold below which a launch takes the 9-limb (8 lanes per integer) layout, MPE_WIDE_DIV: the layout is
# used when WIDE_DIV * batch <= resident groups; (2) ladder waves per CU, MPE_WAVES_PER_CU (8 = two per SIMD, 4 = one per SIMD).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablanes
one() {  # label, env assignments, bench args
  local label=$1 envs=$2; shift 2
  env $envs python bench.py --no-cpu-baseline --no-configs --warmup 1 --dump-launches "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
heavy=[x for x in d['launches_timed_region'] if x['kind'] in (0,3,4,6) and x['ms']>1.0]
n=len(heavy)//d['steps']
print(json.dumps({'case':'$label','env':'$envs','signatures_per_s':round(d['value'],1),'ms_per_step':round(d['ms_per_step'],2),'dominant_frac':round(d['roofline']['frac'],4),'whole_step_frac':round(d['whole_step']['frac'],4),
 'launches_last_step':[[x['kind'],x['bits'],x['exp_words'],x['batch'],x['ms']] for x in heavy[-n:]]}))" | tee -a gpurun_out/ablanes/ab.jsonl
}
for w in 2 3 4 8; do
  one t1n3_1024 "MPE_WIDE_DIV=$w" --sessions 1024 --steps 6
done
for w in 2 4; do
  one t1n3_2048 "MPE_WIDE_DIV=$w" --sessions 2048 --steps 4
  one t1n3_512 "MPE_WIDE_DIV=$w" --sessions 512 --steps 6
done
for v in 8 4; do
  one headline_65536 "MPE_WAVES_PER_CU=$v" --steps 1
  one t1n3_12288 "MPE_WAVES_PER_CU=$v" --sessions 12288 --steps 2
done
