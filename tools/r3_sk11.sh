python -m pytest tests/test_gpu_round3.py -x -q -k "conv_sk" 2>&1 | tail -3
python tools/bench_sk.py --shapes 0 1 2 3 4 --kg 2 8 --tiles 1 2 3 4 --splits 1 2 4 8 2>&1 | grep -v amdgpu.ids
python tools/bench_sk.py --shapes 9 10 11 13 --kg 1 2 8 --tiles 1 2 3 --splits 1 2 4 2>&1 | grep -v amdgpu.ids
