# refresh of the geometry-side evidence only (GPU box): P3b tests, nearest bench, stage times, per-kernel stats of the nearest workload and of the HPR
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_round2.py -m gpu -q 2>&1 | tail -3 > $O/tests.log
python bench.py --workload nearest --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2> $O/benchn.err | tail -1 > $O/bench_nearest.json
python bench.py --workload nearest --shapes-per-step 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2> $O/benchn1.err | tail -1 > $O/bench_nearest_1shape.json
python tools/time_stages.py > $O/stage_times.log 2>&1; cp gpurun_out/stage_times.json $O/stage_times.json
bash tools/prof_nearest.sh > /dev/null 2>&1; cp gpurun_out/kernel_stats_nearest.md $O/kernel_stats_nearest.md
bash tools/prof_hpr.sh > /dev/null 2>&1; cp gpurun_out/kernel_stats_hpr.md $O/kernel_stats_hpr.md; grep -v amdgpu.ids gpurun_out/prof_hpr.log | grep 'ms\|mismatch\|identity' > $O/hpr_dbg.log
cat $O/tests.log; cut -c1-300 $O/bench_nearest.json; cut -c1-300 $O/bench_nearest_1shape.json; cat $O/hpr_dbg.log
