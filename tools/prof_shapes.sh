set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT"/gpurun_out/prof_sh
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/gpurun_out/prof_sh -- python "$GRAFT_REPO_ROOT"/tools/time_shapes.py --iters 20 > "$GRAFT_REPO_ROOT"/gpurun_out/prof_sh.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocpd_stats.py gpurun_out/prof_sh/*/*.db gpurun_out/kernel_stats_shapes.md > /dev/null 2>&1
rm -rf gpurun_out/prof_sh
python tools/time_shapes.py; python tools/time_shapes.py --hpr 0; python tools/time_shapes.py --shapes 4
