cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > gpurun_out/r06/t_rr.log 2>&1; grep -n "passed\|failed" gpurun_out/r06/t_rr.log | tail -2
timeout 900 python tools/bench_rr.py --shapes 3 5 7 8 10 --variants 0 --slabs 0 --no-gn > gpurun_out/r06/bench_rr_dbw.txt 2>&1; cat gpurun_out/r06/bench_rr_dbw.txt
for rr in 0 1 0 1; do
python tools/time_unet.py --rr $rr --batches 1 2 --iters 20 --sampler-steps 20 --out gpurun_out/r06/unet_latency_rr$rr.json > gpurun_out/r06/unet_latency_rr$rr.log 2>&1
echo "rr=$rr"; grep -h batch gpurun_out/r06/unet_latency_rr$rr.log | tail -2 | cut -c1-120
done > gpurun_out/r06/rr_ab2.txt 2>&1; cat gpurun_out/r06/rr_ab2.txt
