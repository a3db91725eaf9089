for sk in 1 2; do echo sk=$sk; python tools/time_unet.py --batches 8 32 --iters 10 --sampler-steps 0 --sk $sk --out gpurun_out/lat_sk$sk.json 2>&1 | grep batch; done
