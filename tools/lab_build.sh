#!/bin/bash
# lab build of libpdhip.so with extra -D flags for the conv kernels: tools/lab_build.sh NAME -DFLAG...  -> pointdreamer_amd/csrc/build/lab_NAME.so
set -e
cd "$(dirname "$0")/../pointdreamer_amd/csrc"
name=$1; shift
for f in nn_gemm nn_conv_halo; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DPD_LAB_BUILD "$@" -c $f.hip -o build/lab_${name}_$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lab_$name.so $(ls build/*.o | grep -v "nn_gemm\|nn_conv_halo\|lab_") build/lab_${name}_nn_gemm.o build/lab_${name}_nn_conv_halo.o
echo built build/lab_$name.so
