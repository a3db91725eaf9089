set -x
cd $GRAFT_REPO_ROOT
python tools/time_unet.py > gpurun_out/unet_latency.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_n1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_n1 -- python $GRAFT_REPO_ROOT/tools/time_unet.py --batches 1 --iters 10 --sampler-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_n1.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_n1/*/*.db > gpurun_out/n1_kernel_stats.md 2>&1
rm -rf gpurun_out/prof_n1
cat gpurun_out/unet_latency.log
