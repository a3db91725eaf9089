python tools/bench_sk.py --shapes 2 3 4 9 10 11 --stages 2 3 4 6 --tiles 1 2 3 --splits 1 2 4 2>&1 | grep -v amdgpu.ids
