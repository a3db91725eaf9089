#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B=pointdreamer_amd/csrc/build
SH="--shapes 4 3 11 --tiles 0 --splits 0"
{
for v in nodma nocompute nobar; do
  echo "=== $v (4 loaders)"; python tools/bench_sk.py --lib $B/lab_$v.so $SH
  echo "=== $v (8 loaders)"; python tools/bench_sk.py --lib $B/lab_$v.so $SH --kg 12
done
echo "=== product"; python tools/bench_sk.py $SH
} > gpurun_out/r05_sk_ablate.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_sk_ablate.txt
bash tools/pmc_sk.sh gpurun_out/r05_pmc_sk_after_reads.json "--shapes 4 --tiles 0 --splits 0 --iters 10" "--shapes 3 --tiles 0 --splits 0 --iters 10" "--shapes 11 --tiles 0 --splits 0 --iters 10" > gpurun_out/pmc_sk.log 2>&1
bash tools/pmc_sk.sh gpurun_out/r05_pmc_sk_base.json "--lib $B/lab_base.so --shapes 4 --tiles 0 --splits 0 --iters 10" "--lib $B/lab_base.so --shapes 3 --tiles 0 --splits 0 --iters 10" "--lib $B/lab_base.so --shapes 11 --tiles 0 --splits 0 --iters 10" > gpurun_out/pmc_sk_base.log 2>&1
rm -rf gpurun_out/pmc_sk
