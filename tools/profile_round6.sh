#!/bin/bash
# Round-6 evidence in ONE gpurun call:
#   1. rocprofv3 --kernel-trace --stats of the headline bench command (dominant kernel's average must agree with the line's HIP events)
#   2. PMC passes (kernel-trace only, one counter group per pass): FETCH_SIZE | WRITE_SIZE | SQ activity -> pmc_traffic.json of THIS commit
#   3. a kernel-trace timeline of ONE lone 1 024-session batch (BASELINE config 4's literal shape), queues / streams included
#   4. bench lines under torch.distributed.run with one rank (session / party through mpe_comm_*), and bench.py's own N>1 command with
#      8 ranks time-sharing the one device of a test box (gloo) incl. the party-sharded mode_b pass, and --config5 at 2 ranks
#   5. the per-wave trace of the ladder kernel with the shipped scheduler (tools/trace_waves.py, profiling build tools/ab/wavetrace.so)
# Usage: tools/profile_round6.sh <tag>     -> gpurun_out/<tag>/
set -u
TAG=${1:-r06prof}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py"
LIGHT="--no-cpu-baseline --no-configs --steps 1"

stats() {  # name, bench args...
  local name=$1; shift
  rm -rf /tmp/p_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o s -- $BENCH $LIGHT --warmup 1 "$@" > "$OUT/stats_${name}_bench.json" 2> /dev/null
  cp /tmp/p_$name/s_kernel_stats.csv "$OUT/kernel_stats_$name.csv" 2> /dev/null
  head -4 "$OUT/kernel_stats_$name.csv"
}
stats headline
stats c5 --t 2 --n 5 --sessions 8192

pmc_pass() {  # name, counters (one string), bench args...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/p_$name
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/p_$name -o c -- $BENCH $LIGHT --warmup 0 "$@" \
    > "$OUT/pmc_${name}_bench.json" 2> "$OUT/pmc_${name}.err"
  python "$ROOT/tools/pmc_summary.py" /tmp/p_$name/c_counter_collection.csv /tmp/p_$name/c_kernel_trace.csv > "$OUT/pmc_$name.json" 2>> "$OUT/pmc_${name}.err"
  tail -c 300 "$OUT/pmc_$name.json"; echo
}
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE
pmc_pass sq "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
MPE_COMMIT=${MPE_COMMIT:-} python "$ROOT/tools/pmc_traffic.py" "$OUT" 65536 > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
tail -c 300 "$OUT/pmc_traffic.json"; echo

rm -rf /tmp/p_lone
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_lone -o t -- $BENCH --sessions 1024 --no-cpu-baseline --no-configs --steps 3 --warmup 1 \
  > "$OUT/lone_1024_traced_bench.json" 2> /dev/null
python "$ROOT/tools/trace_timeline.py" /tmp/p_lone/t_kernel_trace.csv > "$OUT/timeline_lone_1024.json" 2> "$OUT/timeline_lone_1024.err"
head -c 400 "$OUT/timeline_lone_1024.json"; echo

for mode in session party; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    $ROOT/bench.py --gpus 1 --mode $mode $LIGHT --warmup 1 > "$OUT/bench_torchrun1_$mode.json" 2> "$OUT/bench_torchrun1_$mode.err"
  tail -c 200 "$OUT/bench_torchrun1_$mode.json"; echo
done
timeout 1500 $BENCH --gpus 8 --share-device --mode session --sessions 4096 --mode-b-sessions 512 --steps 1 --warmup 1 --no-configs --no-cpu-baseline \
  > "$OUT/bench_8ranks_shared_device_session.json" 2> "$OUT/bench_8ranks_shared_device_session.err"
tail -c 300 "$OUT/bench_8ranks_shared_device_session.json"; echo
timeout 900 $BENCH --gpus 2 --share-device --config5 --sessions 1024 --steps 1 --warmup 1 --no-configs --no-cpu-baseline \
  > "$OUT/bench_2ranks_shared_device_config5.json" 2> "$OUT/bench_2ranks_shared_device_config5.err"
tail -c 300 "$OUT/bench_2ranks_shared_device_config5.json"; echo

if [ -f "$ROOT/tools/ab/wavetrace.so" ]; then
  MPE_LIB_PATH=$ROOT/tools/ab/wavetrace.so python $ROOT/tools/trace_waves.py --cases 16384x2,8192x2,49152x2,655360x2 --reps 2 --sessions 1024 --steps 2 \
    > "$OUT/wave_trace_shipped.jsonl" 2> "$OUT/wave_trace_shipped.err"
  wc -l "$OUT/wave_trace_shipped.jsonl"
fi
ls -la "$OUT"
