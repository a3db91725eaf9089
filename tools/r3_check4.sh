python tools/time_unet.py --batches 1 8 32 --iters 10 --sampler-steps 0 2>&1 | grep batch
python tools/time_unet.py --batches 32 --iters 10 --sampler-steps 0 --sk 0 --fin 0 --out gpurun_out/lat_old.json 2>&1 | grep batch
