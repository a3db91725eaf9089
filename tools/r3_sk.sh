python tools/bench_sk.py --shapes 0 1 2 3 4 --stages 0 2>&1 | grep -v amdgpu.ids
python tools/bench_sk.py --shapes 0 1 --stages 2 4 6 8 --tiles 3 4 --splits 1 4 8 16 32 2>&1 | grep -v amdgpu.ids
python tools/bench_sk.py --shapes 0 1 --tiles 3 4 --splits 4 8 16 32 --fixed 2>&1 | grep -v amdgpu.ids
python tools/bench_sk.py --shapes 5 6 7 8 --stages 0 2>&1 | grep -v amdgpu.ids
