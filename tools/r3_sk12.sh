python -m pytest tests/test_gpu_round3.py -x -q -k "conv_sk or routing" 2>&1 | tail -2
for o in 2 1; do echo "=== order $o"; python tools/bench_sk.py --order $o --shapes 1 9 10 13 14 15 --kg 0 --tiles 1 2 3 --splits 1 2 4 2>&1 | grep -v amdgpu.ids; done
python tools/time_unet.py --batches 1 2 4 8 32 --iters 10 --sampler-steps 0 2>&1 | grep batch
