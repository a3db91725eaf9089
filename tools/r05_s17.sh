#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python tools/bench_conv.py --cfg 0x0x0 --iters 5 --custom 32 256 256 256 256 9 32 256 256 512 256 9 32 128 128 256 256 9 2>&1 | tail -3
python tools/pmc_run.py gpurun_out/r05_pmc_halo.json --filter k_conv3x3_halo -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --cfg 0x0x0 --iters 3 --custom 32 256 256 256 256 9 32 256 256 512 256 9 > gpurun_out/pmc_halo.log 2>&1; tail -3 gpurun_out/pmc_halo.log
