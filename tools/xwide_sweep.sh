#!/bin/bash
# On the GPU box: signatures/s of small batches against the threshold of the 5-limb (4x lanes) layout (MPE_XWIDE_DIV; 0 = off)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/xw
for B in 256 1024 4096; do
for d in 0 4 8 16; do
  MPE_XWIDE_DIV=$d python bench.py --sessions $B --steps 8 --warmup 2 --no-configs --no-cpu-baseline > gpurun_out/xw/b${B}_d$d.json 2> gpurun_out/xw/b${B}_d$d.err
  python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('gpurun_out/xw/b${B}_d$d.json') if l.startswith('{')][-1])
    print('sessions $B xdiv $d: %8.1f sig/s  %7.2f ms/batch signed %s ossl %s' % (d['value'], d['ms_per_step'], d['all_sessions_signed'], d.get('openssl_verified')))
except Exception as e: print('sessions $B xdiv $d FAILED', e)
"
done
done
