set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
# per-kernel times of the hidden-point removal (tools/hpr_dbg.py): tools/prof_hpr.sh [lib.so]
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT"/gpurun_out/prof_hpr
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/gpurun_out/prof_hpr -- python "$GRAFT_REPO_ROOT"/tools/hpr_dbg.py ${1:+"$GRAFT_REPO_ROOT"/$1} > "$GRAFT_REPO_ROOT"/gpurun_out/prof_hpr.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_stats.py gpurun_out/prof_hpr/*/*.db gpurun_out/kernel_stats_hpr.md > /dev/null 2>&1
rm -rf gpurun_out/prof_hpr
grep -v amdgpu.ids gpurun_out/prof_hpr.log | tail -6
grep "hpr" gpurun_out/kernel_stats_hpr.md | cut -c1-60,108-175
