#!/bin/bash
# lab build of libpdhip.so with extra -D flags for nn_gemm.hip: tools/lab_build.sh NAME -DFLAG...  -> pointdreamer_amd/csrc/build/lab_NAME.so
set -e
cd "$(dirname "$0")/../pointdreamer_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c nn_gemm.hip -o build/lab_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lab_$name.so $(ls build/*.o | grep -v "nn_gemm\|lab_") build/lab_$name.o
echo built build/lab_$name.so
