set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT"/gpurun_out/prof_oc
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/gpurun_out/prof_oc -- python "$GRAFT_REPO_ROOT"/tools/time_stages.py > "$GRAFT_REPO_ROOT"/gpurun_out/prof_oc.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_stats.py gpurun_out/prof_oc/*/*.db gpurun_out/kernel_stats_oc.md > /dev/null 2>&1
rm -rf gpurun_out/prof_oc
grep "k_oc\|optcolor" gpurun_out/kernel_stats_oc.md | cut -c1-60,100-180
