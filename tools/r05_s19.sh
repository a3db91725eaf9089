#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for it in 0 8 16; do echo "gn iters $it rep $rep"; python tools/time_unet.py --batches 1 8 32 --iters 5 --sampler-steps 0 --gn-iters $it --out gpurun_out/tmp.json 2>&1 | grep batch; done; done
