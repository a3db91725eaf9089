python tools/bench_sk.py --shapes 3 4 11 13 --kg 8 --stages 2 3 4 --tiles 1 2 --splits 1 2 2>&1 | grep -v amdgpu.ids
