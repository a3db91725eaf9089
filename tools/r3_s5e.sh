set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py -q -x -k "gn_skip or fused_skip" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/s5e_tests.log
cat gpurun_out/s5e_tests.log
timeout 600 python tools/bench_gnskip.py > gpurun_out/s5e_gnskip.txt 2>&1
cat gpurun_out/s5e_gnskip.txt
timeout 600 python tools/time_unet.py --batches 1 8 32 --fskip 0 --out gpurun_out/s5e_lat_off.json > gpurun_out/s5e_lat_off.log 2>&1
timeout 600 python tools/time_unet.py --batches 1 8 32 --fskip 1 --out gpurun_out/s5e_lat_on.json > gpurun_out/s5e_lat_on.log 2>&1
timeout 600 python tools/time_unet.py --batches 1 8 --fskip 2 --out gpurun_out/s5e_lat_all.json > gpurun_out/s5e_lat_all.log 2>&1
cat gpurun_out/s5e_lat_off.log gpurun_out/s5e_lat_on.log gpurun_out/s5e_lat_all.log
