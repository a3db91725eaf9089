#!/bin/bash
# the whole -m gpu suite + smoke, log kept under gpurun_out/ (copied to profiles/r06_gpu_tests.log when it is the round's evidence)
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_tests.log 2>&1; tail -5 gpurun_out/r06_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -3 gpurun_out/r06_smoke.log
