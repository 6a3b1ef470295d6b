#!/bin/bash
# Only the two stall-accounting --pmc passes of tools/profile_round.sh (where a wave of each heavy kernel spends its cycles),
# plus the bench line with the driver's flags.  Usage: tools/profile_stalls.sh <tag>  -> gpurun_out/<tag>/
set -u
TAG=${1:-stalls}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py"
LIGHT="--no-cpu-baseline --no-configs --steps 1"
pmc_pass() {
  local name=$1; shift
  rm -rf /tmp/p_$name
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_$name -o c -- $BENCH $LIGHT --warmup 0 \
    > "$OUT/pmc_${name}_bench.json" 2> "$OUT/pmc_${name}.err"
  python "$ROOT/tools/pmc_summary.py" /tmp/p_$name/c_counter_collection.csv /tmp/p_$name/c_kernel_trace.csv > "$OUT/pmc_$name.json" 2>> "$OUT/pmc_${name}.err"
  tail -c 300 "$OUT/pmc_$name.json"; echo
}
pmc_pass stall_wave_cycles SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pmc_pass stall_inst_counts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVES
timeout 900 $BENCH --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
tail -c 300 "$OUT/bench_driver_flags.json"; echo
