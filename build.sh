#!/bin/bash
# Builds libmpecdsa_hip.so (gfx950) in-tree: three translation units compiled in parallel, then linked.
# Usage: ./build.sh [extra hipcc flags]     (resource usage of every kernel -> build/resource_usage.txt)
set -e
cd "$(dirname "$0")"
mkdir -p build
CS=multi_party_ecdsa_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -save-temps=obj -Rpass-analysis=kernel-resource-usage $*"
pids=()
for u in mpe_lib mpe_pair2048 mpe_pair1024; do
  hipcc $FLAGS -c $CS/$u.hip -o build/$u.o 2> build/resource_usage_$u.txt &
  pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
if [ $fail -ne 0 ]; then grep -h -B2 -A8 "error" build/resource_usage_*.txt | grep -v "^remark" | head -60; echo "BUILD FAILED"; exit 1; fi
hipcc --offload-arch=gfx950 -fPIC -shared -o build/libmpecdsa_hip.so build/mpe_lib.o build/mpe_pair2048.o build/mpe_pair1024.o
cp build/libmpecdsa_hip.so multi_party_ecdsa_amd/libmpecdsa_hip.so
cat build/resource_usage_mpe_lib.txt build/resource_usage_mpe_pair2048.txt build/resource_usage_mpe_pair1024.txt > build/resource_usage.txt
rm -f build/*.bc build/*.hipi build/*.out build/*.hipfb build/*.txt.bak build/*host-x86_64*.s build/*.resolution.txt
grep -E "Function Name|VGPRs:|Occupancy|VGPRs Spill" build/resource_usage.txt | sed 's/remark: [^ ]* *//' | paste - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | awk '{print $3, $5, $6, $9, $10, $13,$14,$15}'
