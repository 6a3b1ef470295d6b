#!/bin/bash
# Re-compiles only the two pair-engine units (seconds) and re-links against the existing build/mpe_lib.o: for iterating on
# mpe_pairexp.h.  A full build is ./build.sh.
set -e
cd "$(dirname "$0")/.."
CS=multi_party_ecdsa_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -save-temps=obj -Rpass-analysis=kernel-resource-usage $*"
for u in mpe_pair2048 mpe_pair1024; do hipcc $FLAGS -c $CS/$u.hip -o build/$u.o 2> build/resource_usage_$u.txt & done
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o build/libmpecdsa_hip.so build/mpe_lib.o build/mpe_pair2048.o build/mpe_pair1024.o
cp build/libmpecdsa_hip.so multi_party_ecdsa_amd/libmpecdsa_hip.so
rm -f build/*.bc build/*.hipi build/*.out build/*.hipfb build/*host-x86_64*.s build/*.resolution.txt
grep -h -A12 "Function Name: .*pair_modexp_kernel" build/resource_usage_mpe_pair2048.txt build/resource_usage_mpe_pair1024.txt | grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill" | sed 's/remark: [^ ]* *//; s/\[-Rpass.*//'
