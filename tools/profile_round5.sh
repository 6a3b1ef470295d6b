#!/bin/bash
# Round-5 evidence in ONE gpurun call (a trimmed tools/profile_round.sh plus what round 5 added):
#   1. bench lines under torch.distributed.run with one rank (session / party: RCCL world 1, the party line through mpe_comm_*) and the
#      driver's own N>1 command on the one device of a test box (2 ranks, gloo)
#   2. rocprofv3 --kernel-trace --stats of the headline bench command and of BASELINE config 5's per-GPU share (t=2 n=5, 8 192 sessions)
#   3. PMC passes (kernel-trace only, one counter group per pass): FETCH_SIZE | WRITE_SIZE | SQ activity for the headline, SQ for config 5
#   4. a kernel-trace timeline of the pipelined engine (a stream of 1 024-session batches, 2 lanes x 4 batches per pass)
# Usage: tools/profile_round5.sh <tag>     -> gpurun_out/<tag>/
set -u
TAG=${1:-r05prof}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py"
LIGHT="--no-cpu-baseline --no-configs --steps 1"

for mode in session party; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    $ROOT/bench.py --gpus 1 --mode $mode $LIGHT --warmup 1 > "$OUT/bench_torchrun1_$mode.json" 2> "$OUT/bench_torchrun1_$mode.err"
  tail -c 200 "$OUT/bench_torchrun1_$mode.json"; echo
done
timeout 900 $BENCH --gpus 2 --share-device --mode session --sessions 16384 --mode-b-sessions 2048 --steps 1 --warmup 1 --no-configs \
  > "$OUT/bench_2ranks_shared_device_session.json" 2> "$OUT/bench_2ranks_shared_device_session.err"
tail -c 200 "$OUT/bench_2ranks_shared_device_session.json"; echo

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
pmc_pass c5_sq "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --t 2 --n 5 --sessions 8192
MPE_COMMIT=${MPE_COMMIT:-} python "$ROOT/tools/pmc_traffic.py" "$OUT" 65536 > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
tail -c 300 "$OUT/pmc_traffic.json"; echo

rm -rf /tmp/p_pipe
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_pipe -o t -- python $ROOT/tools/exp_pipeline.py --lanes 2 --group 4 --batches 24 \
  > "$OUT/pipeline_traced.json" 2> /dev/null
python "$ROOT/tools/trace_streams.py" /tmp/p_pipe/t_kernel_trace.csv > "$OUT/pipeline_streams.json" 2> "$OUT/pipeline_streams.err"
head -c 300 "$OUT/pipeline_streams.json"; echo
ls -la "$OUT"
