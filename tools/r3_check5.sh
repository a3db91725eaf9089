for fin in 0 4 8 32; do echo fin=$fin; python tools/time_unet.py --batches 8 32 --iters 10 --sampler-steps 0 --fin $fin --out gpurun_out/lat_fin$fin.json 2>&1 | grep batch; done
