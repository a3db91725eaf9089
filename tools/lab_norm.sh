#!/bin/bash
# lab build of libpdhip.so with extra -D flags for nn_norm.hip (k_gn_skip variants): tools/lab_norm.sh NAME -DFLAG... -> pointdreamer_amd/csrc/build/lab_NAME.so
# prints the VGPR count and the scratch bytes of k_gn_skip_w1
set -e
cd "$(dirname "$0")/../pointdreamer_amd/csrc"
name=$1; shift
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c nn_norm.hip -o $T/nn_norm.o -save-temps=obj 2>&1 | grep -E "error" || true
grep "k_gn_skip_w1.*num_vgpr\|k_gn_skip_w1.*private_seg_size" $T/nn_norm-hip-amdgcn-amd-amdhsa-gfx950.s | awk '{print $NF}' | tr '\n' ' '
cp $T/nn_norm.o build/lab_${name}_nn_norm.o; rm -rf $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lab_$name.so $(ls build/*.o | grep -v "build/nn_norm.o\|lab_") build/lab_${name}_nn_norm.o -lz
echo built build/lab_$name.so
