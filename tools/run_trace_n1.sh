set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT"/gpurun_out/prof_seq
rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT"/gpurun_out/prof_seq -- python "$GRAFT_REPO_ROOT"/tools/time_unet.py --batches $1 --iters 1 --sampler-steps 3 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/trace_seq.py gpurun_out/prof_seq/*/*.db > gpurun_out/seq_n$1.txt 2>&1
rm -rf gpurun_out/prof_seq
tail -600 gpurun_out/seq_n$1.txt | head -5
