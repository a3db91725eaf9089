python -m pytest tests/test_gpu_geometry.py -x -q -k "linear" 2>&1 | tail -3
python -m pytest tests/test_gpu_nn.py tests/test_gpu_round3.py -x -q 2>&1 | tail -3
python tools/time_stages.py 2>&1 | grep -i "linear\|nearest inpaint"
python tools/time_unet.py --batches 1 8 32 --iters 10 --sampler-steps 10 2>&1 | grep batch
