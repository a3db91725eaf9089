cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_seqn
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_seqn -- python $GRAFT_REPO_ROOT/bench.py --workload nearest --shapes-per-step 1 --steps 5 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/seq_nearest.py gpurun_out/prof_seqn/*/*.db > gpurun_out/s5p_seq.txt 2>&1
rm -rf gpurun_out/prof_seqn
cat gpurun_out/s5p_seq.txt
