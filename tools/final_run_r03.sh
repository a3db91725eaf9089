# round-3 evidence run (GPU box): tests, driver-style bench, side benches, stage / latency tables, rocprofv3 kernel stats, PMC passes
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/tests.log
python bench.py 2> $O/bench.err | tail -1 > $O/bench_full_100steps.json
python bench.py --shapes-per-step 1 --no-cpu-baseline --no-extras 2> $O/bench1.err | tail -1 > $O/bench_full_100steps_1shape.json
python bench.py --workload nearest --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2> $O/benchn.err | tail -1 > $O/bench_nearest.json
python tools/time_stages.py > $O/stage_times.log 2>&1; cp gpurun_out/stage_times.json $O/stage_times.json
python tools/time_unet.py --batches 1 2 4 8 32 > $O/unet_latency.log 2>&1; cp gpurun_out/unet_latency.json $O/unet_latency.json

cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r03
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --ddnm-steps 10 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_r03/*/*.db $O/kernel_stats.md > /dev/null 2>&1
rm -rf gpurun_out/prof_r03
bash tools/prof_nearest.sh > /dev/null 2>&1; cp gpurun_out/kernel_stats_nearest.md $O/kernel_stats_nearest.md
bash tools/prof_hpr.sh > /dev/null 2>&1; cp gpurun_out/kernel_stats_hpr.md $O/kernel_stats_hpr.md; grep -v amdgpu.ids gpurun_out/prof_hpr.log | grep 'ms\|mismatch\|identity' > $O/hpr_dbg.log
rm -rf gpurun_out/pmc_bench
bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bench $O/pmc_conv.json 4 > /dev/null 2>&1
python tools/pmc_kernels.py gpurun_out/pmc_bench $O/pmc_kernels.json > /dev/null 2>&1
rm -rf gpurun_out/pmc_bench/*/*.db
cat $O/tests.log; cut -c1-400 $O/bench_full_100steps.json
