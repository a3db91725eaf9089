#!/bin/bash
# lab build of libpdhip.so with extra -D flags for the small-M conv kernel: tools/lab_sk.sh NAME -DFLAG... -> pointdreamer_amd/csrc/build/labsk_NAME.so
set -e
cd "$(dirname "$0")/../pointdreamer_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c nn_conv_sk.hip -o build/labsk_${name}_nn_conv_sk.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/labsk_$name.so $(ls build/*.o | grep -v "nn_conv_sk\|lab_\|labsk_") build/labsk_${name}_nn_conv_sk.o -lz
echo built build/labsk_$name.so
