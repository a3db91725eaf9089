#!/bin/bash
# round-6 baseline on a GPU box (tree at the start of the round): GPU suite, UNet latency table, batch-1 launch sequence
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
python -m pytest tests -x -q -m gpu > gpurun_out/r06/gpu_tests_baseline.log 2>&1; tail -3 gpurun_out/r06/gpu_tests_baseline.log
python tools/time_unet.py --batches 1 2 4 8 --iters 20 --sampler-steps 20 --out gpurun_out/r06/unet_latency_baseline.json > gpurun_out/r06/unet_latency_baseline.log 2>&1
grep -h batch gpurun_out/r06/unet_latency_baseline.log | tail -8
bash tools/run_trace_n1.sh 1 > /dev/null 2>&1; cp gpurun_out/seq_n1.txt gpurun_out/r06/seq_n1_baseline.txt; head -1 gpurun_out/r06/seq_n1_baseline.txt
