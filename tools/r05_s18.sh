#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python tools/hpr_dbg.py pointdreamer_amd/csrc/build/lab_hprstats.so 2>&1 | grep -v amdgpu.ids
python tools/time_stages.py 2>&1 | grep -v amdgpu.ids | tail -30
