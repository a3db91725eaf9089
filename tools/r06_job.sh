cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -k "conv_rr" 2>&1 | tail -2
timeout 900 python tools/bench_rr.py --taps 1 --gn-mode 1 --shape 8 1024 3072 0 --shape 8 1024 1024 0 --shape 16 1024 3072 0 --shape 16 1024 1024 0 --shape 32 512 1536 0 --shape 32 512 512 0 --variants 0 --slabs 0 1 2 > gpurun_out/r06/bench_rr_1x1.txt 2>&1; cat gpurun_out/r06/bench_rr_1x1.txt
timeout 900 python tools/bench_rr.py --taps 1 --no-gn --no-old --shape 8 1024 3072 0 --shape 8 1024 1024 0 --shape 16 1024 3072 0 --shape 16 1024 1024 0 --shape 32 512 1536 0 --shape 32 512 512 0 --variants 0 --slabs 0 1 2 > gpurun_out/r06/bench_rr_1x1_nogn.txt 2>&1; cat gpurun_out/r06/bench_rr_1x1_nogn.txt
