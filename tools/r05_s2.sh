#!/bin/bash
# round 5, GPU session 2: k_conv_sk -- interleaved fragment reads, eight loader waves: A/B against the round-4 kernel, stamps, parity
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B=pointdreamer_amd/csrc/build
SH="--shapes 4 3 2 1 0 11 10 9 --tiles 0 --splits 0"
{
for rep in 1 2; do
  echo "=== base (round-4 kernel), rep $rep";          python tools/bench_sk.py --lib $B/lab_base.so $SH
  echo "=== reads upfront + new tail reduce, rep $rep"; python tools/bench_sk.py --lib $B/lab_upfront.so $SH
  echo "=== product (interleaved reads), rep $rep";     python tools/bench_sk.py $SH
  echo "=== product, 8 loader waves (kg 12), rep $rep"; python tools/bench_sk.py $SH --kg 12
done
echo "=== product, 8 loaders, stage sweep"; python tools/bench_sk.py --shapes 4 3 11 --tiles 0 --splits 0 --kg 12 --stages 2 3 4
echo "=== product, forced 128x128 tile on the 64^2 layer, kg 8 / 12"; python tools/bench_sk.py --shapes 3 --tiles 1 2 --splits 1 2 --kg 8 12
} > gpurun_out/r05_sk_ab.txt 2>&1
{
echo "=== stamps: product"; python tools/bench_sk.py --lib $B/lab_stamp.so --shapes 4 3 11 --tiles 0 --splits 0 --stamps
echo "=== stamps: 8 loaders"; python tools/bench_sk.py --lib $B/lab_stamp.so --shapes 4 3 11 --tiles 0 --splits 0 --stamps --kg 12
} > gpurun_out/r05_sk_stamps.txt 2>&1
python -m pytest tests/test_gpu_round3.py -x -q -k "conv_sk_small" > gpurun_out/r05_sk_tests.txt 2>&1
tail -3 gpurun_out/r05_sk_tests.txt
cat gpurun_out/r05_sk_ab.txt | grep -v "amdgpu.ids"
