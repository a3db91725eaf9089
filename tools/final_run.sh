set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/final_tests.log
python bench.py 2> gpurun_out/final_bench.err | tail -1 > gpurun_out/final_bench.json
python bench.py --shapes-per-step 1 --no-cpu-baseline 2> gpurun_out/final_bench1.err | tail -1 > gpurun_out/final_bench_1shape.json
python bench.py --workload nearest --steps 20 --warmup 3 --no-cpu-baseline 2> gpurun_out/final_benchn.err | tail -1 > gpurun_out/final_bench_nearest.json
python tools/time_stages.py > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_final
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --ddnm-steps 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_final/*/*.db > gpurun_out/final_kernel_stats.md 2>&1
rm -rf gpurun_out/pmc_bench
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bench gpurun_out/final_pmc.json 4 > /dev/null 2>&1
python tools/pmc_kernels.py gpurun_out/pmc_bench gpurun_out/final_pmc_kernels.json > /dev/null 2>&1
rm -rf gpurun_out/prof_final gpurun_out/pmc_bench/*/*.db
cat gpurun_out/final_tests.log; cat gpurun_out/final_bench.json | cut -c1-600
