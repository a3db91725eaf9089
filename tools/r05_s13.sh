#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python tools/time_demo.py 2>&1 | grep -v "amdgpu.ids\| INFO \|WARNING" > gpurun_out/r05_time_demo.txt; cat gpurun_out/r05_time_demo.txt
