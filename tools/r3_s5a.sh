# session 5, call A: launch-order traces of one DDNM step at batch 1 and 8 + GPU tests on HEAD
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/run_trace_n1.sh 1
bash tools/run_trace_n1.sh 8
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/s5a_tests.log
cat gpurun_out/s5a_tests.log
