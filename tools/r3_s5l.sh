cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/time_neighbor.py > gpurun_out/s5l_neighbor.txt 2>&1
cat gpurun_out/s5l_neighbor.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_nb
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_nb -- python $GRAFT_REPO_ROOT/tools/time_neighbor.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_nb/*/*.db gpurun_out/s5l_kernel_stats_nb.md > /dev/null 2>&1
rm -rf gpurun_out/prof_nb
head -30 gpurun_out/s5l_kernel_stats_nb.md | cut -c1-70,120-190
