#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_round5.py -x -q -k "fp16_unet" 2>&1 | tail -8; grep ddnm_full_fp16 gpurun_out/r05_u1_measured.jsonl | tail -2
