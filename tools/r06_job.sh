cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_round3.py -x -q > gpurun_out/r06/t_rr.log 2>&1; tail -4 gpurun_out/r06/t_rr.log
for g in 0 8 0 8 16; do
python tools/time_unet.py --rr-gn $g --batches 1 2 --iters 20 --sampler-steps 20 --out gpurun_out/r06/unet_latency_g$g.json > gpurun_out/r06/unet_latency_g$g.log 2>&1
echo "rr_gn=$g"; grep -h batch gpurun_out/r06/unet_latency_g$g.log | tail -2 | cut -c1-120
done > gpurun_out/r06/rr_gn_ab.txt 2>&1; cat gpurun_out/r06/rr_gn_ab.txt
