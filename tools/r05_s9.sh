#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/r05_u1_measured.jsonl
python -m pytest tests/test_gpu_round5.py tests/test_gpu_nn.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -k "unet or skip or checkpoint" > gpurun_out/r05_tests_u1.txt 2>&1; tail -15 gpurun_out/r05_tests_u1.txt
