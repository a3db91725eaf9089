#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B=pointdreamer_amd/csrc/build
SH="--no-old --tiles 0 --splits 0 --shape 1 128 128 256 256 9 --shape 1 128 128 1024 256 9 --shape 1 64 64 512 512 9 --shape 8 32 32 512 512 9 --shape 8 16 16 1024 1024 9 --shape 1 32 32 512 512 9"
{
for rep in 1 2; do
echo "=== product rep $rep"; python tools/bench_sk.py $SH
for v in cm cm_nt nt sc1 cm_sc1; do echo "=== $v rep $rep"; python tools/bench_sk.py --lib $B/lab_$v.so $SH; done
done
for v in nocompute nc_cm nc_nt nc_cm_nt; do echo "=== $v"; python tools/bench_sk.py --lib $B/lab_$v.so $SH; done
} > gpurun_out/r05_sk_order.txt 2>&1
grep -v "amdgpu.ids\|igemm" gpurun_out/r05_sk_order.txt | paste - - - - - - - | sed 's/  */ /g; s/kg 0 stages 0 tile 0: s0://g'
