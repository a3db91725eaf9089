#!/bin/bash
# round 5, GPU session 1: LDS micro-benchmark, k_conv_sk stamps / variants / PMC, start-of-round UNet latency
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B=pointdreamer_amd/csrc/build
tools/ub/ub_lds.bin > gpurun_out/ub_lds.txt 2>&1; cat gpurun_out/ub_lds.txt
for v in stamp stamppin stampnodma; do
  echo "=== $v" ; python tools/bench_sk.py --lib $B/lab_$v.so --shapes 4 3 11 --tiles 0 --splits 0 --stamps
done > gpurun_out/sk_stamps.txt 2>&1
tail -60 gpurun_out/sk_stamps.txt
for rep in 1 2; do
  for v in product pin; do
    echo "=== $v (rep $rep)"; if [ $v = product ]; then python tools/bench_sk.py --shapes 4 3 2 1 11 10 --tiles 0 --splits 0; else python tools/bench_sk.py --lib $B/lab_$v.so --shapes 4 3 2 1 11 10 --tiles 0 --splits 0; fi
  done
done > gpurun_out/sk_ab_pin.txt 2>&1
cat gpurun_out/sk_ab_pin.txt
(cd /tmp && rocprofv3 -L > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_counters.txt" 2>&1)
bash tools/pmc_sk.sh gpurun_out/r05_pmc_sk.json "--shapes 4 --tiles 0 --splits 0 --iters 10" "--shapes 3 --tiles 0 --splits 0 --iters 10" "--shapes 11 --tiles 0 --splits 0 --iters 10" > gpurun_out/pmc_sk.log 2>&1
tail -5 gpurun_out/pmc_sk.log
python tools/time_unet.py --batches 1 8 --out gpurun_out/r05_unet_latency_start.json > gpurun_out/unet_latency_start.log 2>&1; tail -5 gpurun_out/unet_latency_start.log
