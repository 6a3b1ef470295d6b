#!/bin/bash
# signatures/s and ms per batch against the batch size (one box): gpurun_out/sweep/batch_sweep.jsonl
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/sweep
: > gpurun_out/sweep/batch_sweep.jsonl
run() {  # t n sessions steps
  python bench.py --t $1 --n $2 --sessions $3 --steps $4 --warmup 1 --no-configs --no-cpu-baseline 2> gpurun_out/sweep/err_$1_$2_$3.txt | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'t':$1,'n':$2,'sessions':$3,'signatures_per_s':round(d['value'],1),'ms_per_batch':round(d['ms_per_step'],2),'all_signed':d['all_sessions_signed'],'openssl_verified':d.get('openssl_verified'),'dominant_kernel_frac':d['roofline']['frac'],'whole_step_frac':d['whole_step']['frac']}))
" | tee -a gpurun_out/sweep/batch_sweep.jsonl
}
for B in 64 256 1024 4096 16384; do run 1 3 $B 6; done
run 1 3 65536 3
run 1 3 131072 2
for B in 1024 8192 16384; do run 2 5 $B 3; done
