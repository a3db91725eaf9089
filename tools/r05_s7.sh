#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B=pointdreamer_amd/csrc/build
python -m pytest tests/test_gpu_round3.py tests/test_gpu_nn.py tests/test_gpu_round4.py -x -q -k "conv_sk or unet or skip or ddnm" > gpurun_out/r05_tests_sk.txt 2>&1; tail -4 gpurun_out/r05_tests_sk.txt
for rep in 1 2; do
PDHIP_LAB_LIB=$B/lab_base.so python tools/time_unet.py --batches 1 8 --out gpurun_out/r05_lat_base_$rep.json 2>&1 | grep batch
python tools/time_unet.py --batches 1 8 --out gpurun_out/r05_lat_new_$rep.json 2>&1 | grep batch
done
