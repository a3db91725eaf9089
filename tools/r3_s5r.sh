cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -x -k "not unet and not ddnm and not attention and not conv" 2>&1 | grep -E "passed|failed|rror" | tail -4 > gpurun_out/s5r_tests.log
cat gpurun_out/s5r_tests.log
