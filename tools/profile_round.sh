#!/bin/bash
# Collects the evidence bench.py's roofline line refers to, on a GPU box:
#   1. the default bench line (cpu_baseline, every BASELINE config)
#   2. the same bench under torch.distributed.run with one rank, session-sharded and party-sharded (the N>1 code paths:
#      RCCL init, barriers, max-reduce; in party mode the per-round all-gather driver)
#   3. rocprofv3 --kernel-trace --stats of the bench command (per-kernel durations) + the timeline of a 1 024-session step
#   4. separate rocprofv3 --pmc passes (kernel-trace only): FETCH_SIZE | WRITE_SIZE | SQ/GRBM activity
# Usage (from the repo root):  tools/profile_round.sh <tag> [quick]     -> gpurun_out/<tag>/...
# tools/pmc_summary.py turns the counter CSVs into profiles/<round>/pmc_*.json.
set -u
TAG=${1:-prof}
QUICK=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py"
LIGHT="--no-cpu-baseline --no-configs --steps 1"

timeout 900 $BENCH > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 300 "$OUT/bench_default.json"; echo

for mode in session party; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    $ROOT/bench.py --gpus 1 --mode $mode $LIGHT --warmup 1 > "$OUT/bench_torchrun1_$mode.json" 2> "$OUT/bench_torchrun1_$mode.err"
  tail -c 200 "$OUT/bench_torchrun1_$mode.json"; echo
done

# the N>1 launch as the driver starts it, on the one GPU a test box has: bench.py spawns its own ranks, every rank on cuda:0,
# the round slabs through a gloo all-gather (a functional run of the code path; the ranks time-share the device)
# (round 4: the session-mode line carries the all-gather layout self-test, a per-rank parity sample against the oracle and the
#  party-sharded pass at BASELINE config 5's shape — mode_b{} — so it runs WITH the oracle legs)
for mode in session party; do
  timeout 900 $BENCH --gpus 2 --share-device --mode $mode --sessions 16384 --mode-b-sessions 2048 --steps 1 --warmup 1 --no-configs \
    > "$OUT/bench_2ranks_shared_device_$mode.json" 2> "$OUT/bench_2ranks_shared_device_$mode.err"
  tail -c 200 "$OUT/bench_2ranks_shared_device_$mode.json"; echo
done
MPE_FB_WINDOW_BITS=10 timeout 900 $BENCH --gpus 8 --share-device --sessions 512 --mode-b-sessions 256 --steps 1 --warmup 1 --no-configs \
  > "$OUT/bench_8ranks_shared_device_session.json" 2> "$OUT/bench_8ranks_shared_device_session.err"
tail -c 200 "$OUT/bench_8ranks_shared_device_session.json"; echo

rm -rf /tmp/p_stats /tmp/p_1k
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- $BENCH $LIGHT --warmup 1 \
  > "$OUT/stats_bench.json" 2> /dev/null
cp /tmp/p_stats/s_kernel_stats.csv "$OUT/kernel_stats.csv" 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_1k -o t -- $BENCH --sessions 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-configs \
  > "$OUT/bench_1024_traced.json" 2> /dev/null
python "$ROOT/tools/trace_timeline.py" /tmp/p_1k/t_kernel_trace.csv > "$OUT/timeline_1024.json" 2> "$OUT/timeline_1024.err"
head -c 200 "$OUT/timeline_1024.json"; echo
[ -n "$QUICK" ] && { ls -la "$OUT"; exit 0; }

pmc_pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/p_$name
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_$name -o c -- $BENCH $LIGHT --warmup 0 \
    > "$OUT/pmc_${name}_bench.json" 2> "$OUT/pmc_${name}.err"
  python "$ROOT/tools/pmc_summary.py" /tmp/p_$name/c_counter_collection.csv /tmp/p_$name/c_kernel_trace.csv > "$OUT/pmc_$name.json" 2>> "$OUT/pmc_${name}.err"
  tail -c 400 "$OUT/pmc_$name.json"; echo
}
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE
pmc_pass sq GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
# where a wave spends its cycles (two more passes: the SQ block counts 8 counters at a time)
pmc_pass stall_wave_cycles SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pmc_pass stall_inst_counts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVES
ls -la "$OUT"
