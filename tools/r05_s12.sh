#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for it in 0 16 8 4 2 1 0; do echo "gn iters $it"; python tools/time_unet.py --batches 32 --iters 5 --sampler-steps 0 --gn-iters $it --out gpurun_out/tmp.json 2>&1 | grep batch; done
