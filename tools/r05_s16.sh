#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_demo.py tests/test_gpu_round2.py -x -q -k "demo or cli or files or png or batched or directory" 2>&1 | tail -3
python tools/time_demo.py --configs nearest 2>&1 | grep -v "amdgpu.ids\| INFO \|WARNING" > gpurun_out/r05_time_demo_n.txt; cat gpurun_out/r05_time_demo_n.txt
python tools/prof_demo_dir.py 2>&1 | grep -v "amdgpu.ids\| INFO \|WARNING" | cut -c1-170 | grep -A28 "^wall"
