#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B=pointdreamer_amd/csrc/build
tools/ub/ub_lds.bin > gpurun_out/r05_ub_lds.txt 2>&1
python tools/bench_sk.py --lib $B/lab_stamp.so --no-old --shapes 4 3 11 --tiles 0 --splits 0 --stamps 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_sk_loop_stamps.txt
cat gpurun_out/r05_ub_lds.txt | tail -4; cut -c1-250 gpurun_out/r05_sk_loop_stamps.txt
