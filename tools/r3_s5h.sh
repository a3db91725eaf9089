cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py -q -x -k "refine" 2>&1 | tail -30 > gpurun_out/s5h_tests.log
cat gpurun_out/s5h_tests.log
