cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q > gpurun_out/r06/t_rr.log 2>&1; tail -4 gpurun_out/r06/t_rr.log
timeout 1200 python tools/bench_rr.py --variants 0 5 6 7 --slabs 1 2 4 8 --no-gn > gpurun_out/r06/bench_rr_v4.txt 2>&1; cat gpurun_out/r06/bench_rr_v4.txt
timeout 1200 python tools/bench_rr.py --shapes 0 3 7 --variants 0 6 7 --slabs 0 1 2 4 --no-gn --batches 2 4 8 > gpurun_out/r06/bench_rr_v4b.txt 2>&1; cat gpurun_out/r06/bench_rr_v4b.txt
