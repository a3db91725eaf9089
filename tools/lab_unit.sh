#!/bin/bash
# lab build of libpdhip.so with extra -D flags for ONE translation unit: tools/lab_unit.sh NAME UNIT -DFLAG... -> csrc/build/lab_NAME.so
set -e
cd "$(dirname "$0")/../pointdreamer_amd/csrc"
name=$1; unit=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DPD_LAB_BUILD "$@" -c $unit.hip -o build/lab_${name}_$unit.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lab_$name.so $(ls build/*.o | grep -v "build/$unit.o\|lab_\|labsk_") build/lab_${name}_$unit.o -lz
echo built build/lab_$name.so
