cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python bench.py --steps 1 --warmup 0 --ddnm-steps 10 --no-cpu-baseline > gpurun_out/r06/bench_short.json 2> gpurun_out/r06/bench_short.err; tail -3 gpurun_out/r06/bench_short.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r06/bench_short.json') if l.startswith('{')][-1]); print(json.dumps(d['extras'].get('view_parallel_projection')), d['extras']['nearest_stacked'])"
