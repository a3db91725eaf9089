python -m pytest tests/test_gpu_round3.py -x -q -k "conv_sk or routing" 2>&1 | tail -3
python tools/bench_sk.py --shapes 12 13 14 9 --kg 1 2 --tiles 1 2 3 --splits 1 2 4 2>&1 | grep -v amdgpu.ids
