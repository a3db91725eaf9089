#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for on in 0 1; do echo "skgn $on"; python tools/time_unet.py --batches 1 2 8 32 --skgn $on --out gpurun_out/tmp.json 2>&1 | grep batch; done; done
