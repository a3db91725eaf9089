#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B=pointdreamer_amd/csrc/build
SH="--no-old --tiles 1 --splits 1 --shape 1 128 128 256 256 9 --shape 1 128 128 1024 256 9 --shape 1 64 64 1024 1024 9"
{
for v in nocompute nc_noa nc_nob nc_noa_rot nc_nobar; do
  for kg in 8 12; do for st in 3 4; do echo "=== $v kg $kg stages $st"; python tools/bench_sk.py --lib $B/lab_$v.so $SH --kg $kg --stages $st; done; done
done
} > gpurun_out/r05_sk_fill.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_sk_fill.txt
(cd /tmp && rocprofv3 -L > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_counters.txt" 2>&1); wc -l gpurun_out/rocprof_counters.txt
