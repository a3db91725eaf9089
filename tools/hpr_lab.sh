# libpdhip.so + the -DPD_HPR_STATS lab build of the hidden-point removal (csrc/build/lab_hprstats.so) for tools/hpr_dbg.py
set -e
cd "$(dirname "$0")/../pointdreamer_amd/csrc"
make 2>&1 | grep -i "error\|warning" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DPD_HPR_STATS "$@" -c hpr.hip -o build/lab_stats_hpr.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lab_hprstats.so $(ls build/*.o | grep -v "lab_\|build/hpr.o") build/lab_stats_hpr.o -lz
