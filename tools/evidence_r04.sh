#!/bin/bash
# Round-4 evidence set (GPU box): bench line, rocprofv3 kernel stats of the same command, PMC passes, UNet latency table, stage times.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ev
python bench.py > gpurun_out/ev/bench_full.json 2> gpurun_out/ev/bench_full.err
python tools/time_unet.py --batches 1 2 4 8 32 --iters 20 --sampler-steps 20 --out gpurun_out/ev/unet_latency.json > gpurun_out/ev/unet_latency.log 2>&1
python tools/time_stages.py > gpurun_out/ev/stage_times.log 2>&1; cp gpurun_out/stage_times.json gpurun_out/ev/
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT"/gpurun_out/ev/prof
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/gpurun_out/ev/prof -- python "$GRAFT_REPO_ROOT"/bench.py --steps 1 --warmup 0 --ddnm-steps 10 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT"/gpurun_out/ev/prof.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_stats.py gpurun_out/ev/prof/*/*.db gpurun_out/ev/kernel_stats.md > /dev/null 2>&1
rm -rf gpurun_out/ev/prof
bash tools/pmc_bench.sh > gpurun_out/ev/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bench gpurun_out/ev/pmc_conv.json > /dev/null 2>&1
python tools/pmc_kernels.py gpurun_out/pmc_bench gpurun_out/ev/pmc_kernels.json > /dev/null 2>&1
rm -rf gpurun_out/pmc_bench
tail -c 600 gpurun_out/ev/bench_full.json; cat gpurun_out/ev/unet_latency.log
