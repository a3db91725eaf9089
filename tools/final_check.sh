set -x
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --workload nearest --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_nearest.json
bash tools/prof_nearest.sh > /dev/null 2>&1
python tools/time_stages.py > /dev/null 2>&1
python -c "
import json; d=json.load(open('gpurun_out/bench_nearest.json')); print('nearest', d['value'], d['ms_per_step'])"
head -8 gpurun_out/kernel_stats_nearest.md | cut -c1-160
