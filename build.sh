#!/bin/bash
# Builds libmpecdsa_hip.so (gfx950) in-tree.  Usage: ./build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
mkdir -p build
SRC="multi_party_ecdsa_amd/csrc/mpe_lib.hip"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -save-temps=obj \
  -Rpass-analysis=kernel-resource-usage "$@" -o build/libmpecdsa_hip.so $SRC 2> build/resource_usage.txt
cp build/libmpecdsa_hip.so multi_party_ecdsa_amd/libmpecdsa_hip.so
rm -f build/*.bc build/*.hipi build/*.o build/*.out build/*.hipfb build/*.txt.bak build/*host-x86_64*.s build/*.resolution.txt
grep -E "Function Name|VGPRs:|Occupancy|VGPRs Spill" build/resource_usage.txt | sed 's/remark: [^ ]* *//' | paste - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | awk '{print $3, $5, $6, $9, $10, $13,$14,$15}'
