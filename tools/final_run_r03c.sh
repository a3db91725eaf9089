# round-3 evidence run, final refresh (GPU box), part A: tests, driver-style bench, side benches, stage / latency tables
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|FAILED" | tail -8 > $O/tests.log
python bench.py 2> $O/bench.err | tail -1 > $O/bench_full_100steps.json
python bench.py --shapes-per-step 1 --no-cpu-baseline --no-extras 2> $O/bench1.err | tail -1 > $O/bench_full_100steps_1shape.json
python bench.py --workload nearest --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2> $O/benchn.err | tail -1 > $O/bench_nearest.json
python tools/time_stages.py > $O/stage_times.log 2>&1; cp gpurun_out/stage_times.json $O/stage_times.json
python tools/time_unet.py --batches 1 2 4 8 32 > $O/unet_latency.log 2>&1; cp gpurun_out/unet_latency.json $O/unet_latency.json
python tools/bench_gnskip.py > $O/gnskip_bench.txt 2>&1
python tools/bench_attn.py > $O/attn_bench.txt 2>&1
python tools/time_demo.py > $O/time_demo.txt 2>&1
cat $O/tests.log; cut -c1-300 $O/bench_full_100steps.json
