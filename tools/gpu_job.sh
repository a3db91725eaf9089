#!/bin/bash
# One parametrised runner for GPU-box jobs (replaces the per-experiment r3_*.sh scripts):
#   gpurun -- 'bash tools/gpu_job.sh <name> "<cmd1>" "<cmd2>" ...'
# runs each command from the repo root and writes its output to gpurun_out/<name>_<i>.log (tail shown at the end).
name=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
i=0
for cmd in "$@"; do
    i=$((i+1))
    echo "== [$name $i] $cmd" | tee gpurun_out/${name}_$i.log
    ( eval "$cmd" ) >> gpurun_out/${name}_$i.log 2>&1
    echo "   rc=$?" | tee -a gpurun_out/${name}_$i.log
    tail -4 gpurun_out/${name}_$i.log
done
