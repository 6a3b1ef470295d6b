#!/bin/bash
# On the GPU box: the super-batch experiment (tools/exp_superbatch.py) over lanes x group, with and without forked streams.
# CONFIGS="lanes:group:nopar ..."   e.g.  CONFIGS="1:4:0 2:4:1" tools/exp_superbatch_sweep.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/superbatch
for c in ${CONFIGS:="1:1:0 1:4:0 1:8:0 2:4:1 2:4:0 3:4:1 2:8:1 3:8:1 4:2:1 4:2:0 2:2:0 3:2:0"}; do
  IFS=: read lanes group nopar <<< "$c"
  reps=$(( 48 / (lanes * group) )); [ $reps -lt 2 ] && reps=2
  if [ "$nopar" = 1 ]; then export MPE_NO_PAR=1; else unset MPE_NO_PAR; fi
  timeout 300 python tools/exp_superbatch.py --lanes $lanes --group $group --reps $reps 2> gpurun_out/superbatch/err_$c.txt | tee -a gpurun_out/superbatch/sweep.jsonl
done
