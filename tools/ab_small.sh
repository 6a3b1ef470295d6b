#!/bin/bash
# One 1 024-session batch (BASELINE config 4's literal shape) under the switches that decide how its launches share the chip:
# MPE_NO_PAR (no forked streams: round 1 takes the merged-ladder path), MPE_WIDE_DIV (9-limb layout threshold), MPE_XWIDE_DIV (5-limb).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/absmall
one() {  # label, env assignments, bench args
  local label=$1 envs=$2; shift 2
  env $envs python bench.py --no-cpu-baseline --no-configs --warmup 1 --dump-launches "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
heavy=[x for x in d['launches_timed_region'] if x['kind'] in (0,3,4,6) and x['ms']>1.0]
n=len(heavy)//d['steps']
print(json.dumps({'case':'$label','env':'$envs','signatures_per_s':round(d['value'],1),'ms_per_step':round(d['ms_per_step'],2),
 'launches_last_step':[[x['kind'],x['bits'],x['exp_words'],x['batch'],x['ms']] for x in heavy[-n:]]}))" | tee -a gpurun_out/absmall/ab.jsonl
}
for e in "MPE_X=0" "MPE_NO_PAR=1" "MPE_NO_PAR=1 MPE_WIDE_DIV=4" "MPE_NO_PAR=1 MPE_WIDE_DIV=3" "MPE_NO_PAR=1 MPE_WIDE_DIV=4 MPE_XWIDE_DIV=8" "MPE_WIDE_DIV=4 MPE_XWIDE_DIV=8"; do
  one t1n3_1024 "$e" --sessions 1024 --steps 6
done
for e in "MPE_X=0" "MPE_NO_PAR=1 MPE_WIDE_DIV=4"; do
  one t1n3_512 "$e" --sessions 512 --steps 6
  one t1n3_2048 "$e" --sessions 2048 --steps 4
done
