cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_geometry.py -q -x -k "optimize_color" 2>&1 | tail -15 > gpurun_out/s5k_tests.log
cat gpurun_out/s5k_tests.log
bash tools/prof_optimize_color.sh > gpurun_out/s5k_prof.txt 2>&1
cat gpurun_out/s5k_prof.txt
grep "optimize_color" -A3 gpurun_out/stage_times.json | head
