# round-3 evidence run, part B (GPU box): rocprofv3 kernel stats of the bench command, PMC passes (separate, --pmc only with --kernel-trace)
set -x
cd /tmp && export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $GRAFT_REPO_ROOT/$O
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r03
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --ddnm-steps 10 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_r03/*/*.db $O/kernel_stats.md > /dev/null 2>&1
rm -rf gpurun_out/prof_r03
bash tools/prof_nearest.sh > /dev/null 2>&1; cp gpurun_out/kernel_stats_nearest.md $O/kernel_stats_nearest.md
rm -rf gpurun_out/pmc_bench
bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bench $O/pmc_conv.json 4 > /dev/null 2>&1
python tools/pmc_kernels.py gpurun_out/pmc_bench $O/pmc_kernels.json > /dev/null 2>&1
rm -rf gpurun_out/pmc_bench/*/*.db gpurun_out/pmc_bench
head -12 $O/kernel_stats.md | cut -c1-150
