cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for q in 1 2 3; do
  for sps in 8; do
    echo "== GPU_MAX_HW_QUEUES=$q shapes-per-step=$sps"
    GPU_MAX_HW_QUEUES=$q python bench.py --workload nearest --shapes-per-step $sps --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step']/$sps)"
  done
done > gpurun_out/s5m.txt 2>&1
cat gpurun_out/s5m.txt
