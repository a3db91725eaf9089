#!/bin/bash
# Round-6 evidence set (GPU box): bench line (defaults), rocprofv3 kernel stats of the same command, PMC passes (dominant kernel + per
# kernel), UNet latency table, stage times, nearest-workload kernel stats.  Everything lands in gpurun_out/ev6/ (small files only).
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ev6; E=gpurun_out/ev6
python bench.py > $E/bench_full.json 2> $E/bench_full.err
python tools/time_unet.py --batches 1 2 4 8 32 --iters 20 --sampler-steps 20 --out $E/unet_latency.json > $E/unet_latency.log 2>&1
python tools/time_stages.py > $E/stage_times.log 2>&1; cp gpurun_out/stage_times.json $E/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT"/$E/prof
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/$E/prof -- python "$GRAFT_REPO_ROOT"/bench.py --steps 1 --warmup 0 --ddnm-steps 10 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT"/$E/prof.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_stats.py $E/prof/*/*.db $E/kernel_stats.md > /dev/null 2>&1
rm -rf $E/prof
bash tools/pmc_bench.sh > $E/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bench $E/pmc_conv.json > /dev/null 2>&1
python tools/pmc_kernels.py gpurun_out/pmc_bench $E/pmc_kernels.json > /dev/null 2>&1
rm -rf gpurun_out/pmc_bench
cd /tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/$E/profn -- python "$GRAFT_REPO_ROOT"/bench.py --workload nearest --shapes-per-step 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT"/$E/profn.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_stats.py $E/profn/*/*.db $E/kernel_stats_nearest.md > /dev/null 2>&1
rm -rf $E/profn
tail -c 900 $E/bench_full.json; cat $E/unet_latency.log | grep batch
