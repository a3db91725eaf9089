#!/bin/bash
# the whole -m gpu suite, log kept under gpurun_out/ (copied to profiles/r05_gpu_tests.log when it is the round's evidence)
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1; tail -5 gpurun_out/r05_gpu_tests.log
