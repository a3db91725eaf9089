#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for on in 0 1; do
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_$on
rocprofv3 --kernel-trace --stats -d /tmp/prof_$on -- python "$GRAFT_REPO_ROOT"/tools/time_unet.py --batches 1 --iters 10 --sampler-steps 0 --skgn $on --out /tmp/x.json > /tmp/prof_$on.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_stats.py /tmp/prof_$on/*/*.db gpurun_out/n1_kernel_stats_skgn$on.md > /dev/null 2>&1
done
head -32 gpurun_out/n1_kernel_stats_skgn0.md | cut -c1-175; echo ======; head -40 gpurun_out/n1_kernel_stats_skgn1.md | cut -c1-175
