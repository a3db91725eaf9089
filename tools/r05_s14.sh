#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
python tools/prof_demo_dir.py 2>&1 | grep -v "amdgpu.ids\| INFO " | cut -c1-170 > gpurun_out/prof_demo_dir.txt; head -80 gpurun_out/prof_demo_dir.txt
