#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B=pointdreamer_amd/csrc/build
for rep in 1 2; do
echo product; python tools/time_unet.py --batches 1 8 --out gpurun_out/tmp.json 2>&1 | grep batch
for v in ntsk ntgn ntboth; do echo $v; PDHIP_LAB_LIB=$B/lab_$v.so python tools/time_unet.py --batches 1 8 --out gpurun_out/tmp.json 2>&1 | grep batch; done
done
