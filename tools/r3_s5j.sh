cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -12 > gpurun_out/s5j_tests.log
cat gpurun_out/s5j_tests.log
