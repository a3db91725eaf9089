#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B=pointdreamer_amd/csrc/build
SH="--no-old --tiles 1 --splits 1 --shape 1 128 128 64 256 9 --shape 1 128 128 128 256 9 --shape 1 128 128 256 256 9 --shape 1 128 128 512 256 9 --shape 1 128 128 1024 256 9 --shape 1 128 128 256 256 1 --shape 1 128 128 1024 256 1"
{
echo "=== product"; python tools/bench_sk.py $SH
for v in nodma nocompute nobar; do
  echo "=== $v"; python tools/bench_sk.py --lib $B/lab_$v.so $SH
done
echo "=== product kg12"; python tools/bench_sk.py $SH --kg 12
echo "=== product kg1 (4 waves: load+compute)"; python tools/bench_sk.py $SH --kg 1
echo "=== product kg2"; python tools/bench_sk.py $SH --kg 2
} > gpurun_out/r05_sk_kslope.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_sk_kslope.txt
