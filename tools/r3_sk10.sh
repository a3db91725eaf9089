python tools/bench_sk.py --shapes 4 --kg 8 2 --stages 3 --tiles 1 --splits 1 --stamps --lib pointdreamer_amd/csrc/build/labsk_stamp.so 2>&1 | grep -v amdgpu.ids
python tools/bench_sk.py --shapes 0 --kg 8 2 --stages 4 --tiles 4 --splits 8 --stamps --lib pointdreamer_amd/csrc/build/labsk_stamp.so 2>&1 | grep -v amdgpu.ids
