python -m pytest tests/test_gpu_round3.py -x -q -k "conv_sk or routing" 2>&1 | tail -15
for sk in 0 1; do for fin in 0 4; do echo "sk=$sk fin=$fin"; python tools/time_unet.py --batches 1 8 --iters 10 --sampler-steps 0 --sk $sk --fin $fin --out gpurun_out/lat_sk${sk}_fin${fin}.json 2>&1 | grep batch; done; done
