# build-side counterpart: tools/hpr_lab.sh; on the GPU box: P3b parity tests, round statistics (lab build), per-kernel times
timeout 600 python -m pytest tests/test_gpu_geometry.py -x -q -m gpu -k "p3b" 2>&1 | tail -5
timeout 200 python tools/hpr_dbg.py pointdreamer_amd/csrc/build/lab_hprstats.so 2>&1 | grep -v amdgpu.ids | tail -7
bash tools/prof_hpr.sh 2>&1 | grep hpr | awk -F"|" '{print substr($2,1,50), $3,$4,$5,$6,$7}'
