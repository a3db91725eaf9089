cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py -q -x -k "gn_skip or optimize_color_is" 2>&1 | tail -15 > gpurun_out/s5s_tests.log
cat gpurun_out/s5s_tests.log
