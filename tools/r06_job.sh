cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for rr in 0 1 0 1; do
python tools/time_unet.py --rr $rr --batches 1 2 4 --iters 20 --sampler-steps 20 --out gpurun_out/r06/unet_latency_rr$rr.json > gpurun_out/r06/unet_latency_rr$rr.log 2>&1
echo "rr=$rr"; grep -h batch gpurun_out/r06/unet_latency_rr$rr.log | tail -3 | cut -c1-120
done > gpurun_out/r06/rr_ab.txt 2>&1; cat gpurun_out/r06/rr_ab.txt
