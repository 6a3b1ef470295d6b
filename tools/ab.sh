#!/bin/bash
# A/B builds of the pair-engine units: tools/ab.sh NAME [extra hipcc flags]  ->  tools/ab/NAME.so (linked against build/mpe_lib.o)
# Run a variant with MPE_LIB_PATH=tools/ab/NAME.so python bench.py ...   (the .so files travel with gpurun, not with git)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/ab /tmp/ab_$NAME
CS=multi_party_ecdsa_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -save-temps=obj -Rpass-analysis=kernel-resource-usage $*"
for u in mpe_pair2048 mpe_pair1024; do hipcc $FLAGS -c $CS/$u.hip -o /tmp/ab_$NAME/$u.o 2> /tmp/ab_$NAME/res_$u.txt & done
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o tools/ab/$NAME.so ${AB_LIB_O:-build/mpe_lib.o} /tmp/ab_$NAME/mpe_pair2048.o /tmp/ab_$NAME/mpe_pair1024.o
grep -h -A12 "Function Name: .*pair_modexp_kernelINS_3CfgILi2048ELi29ELi18ELi4" /tmp/ab_$NAME/res_mpe_pair2048.txt | grep -E "VGPRs:|ScratchSize|VGPRs Spill" | sed 's/remark: [^ ]* *//; s/\[-Rpass.*//' | tr '\n' ' '; echo
