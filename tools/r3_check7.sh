python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | tail -2
python tools/time_unet.py --batches 1 2 4 8 32 --iters 10 --sampler-steps 0 2>&1 | grep batch
