python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | tail -3
python tools/time_unet.py --batches 1 2 4 8 --iters 10 --sampler-steps 0 2>&1 | grep batch
python tools/time_unet.py --batches 1 8 --iters 10 --sampler-steps 0 --sk 0 --fin 0 --out gpurun_out/lat_old.json 2>&1 | grep batch
bash tools/run_trace_both.sh > /dev/null 2>&1
