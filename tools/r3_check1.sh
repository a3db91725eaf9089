python -m pytest tests/test_gpu_round3.py -x -q -s 2>&1 | tail -15
python -m pytest tests/test_gpu_geometry.py -x -q -k "linear" 2>&1 | tail -5
python tools/time_unet.py --batches 1 2 4 8 --iters 10 --sampler-steps 0 2>&1 | tail -5
