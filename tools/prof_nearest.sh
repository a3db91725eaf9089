set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
set -x
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT"/gpurun_out/prof_nearest
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/gpurun_out/prof_nearest -- python "$GRAFT_REPO_ROOT"/bench.py --workload nearest --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT"/gpurun_out/prof_nearest.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_stats.py gpurun_out/prof_nearest/*/*.db gpurun_out/kernel_stats_nearest.md > /dev/null 2>&1
rm -rf gpurun_out/prof_nearest
head -45 gpurun_out/kernel_stats_nearest.md | cut -c1-170
