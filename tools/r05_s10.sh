#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
ls /sys/class/drm/ 2>&1 | head -20; for c in /sys/class/drm/card*/device; do echo $c $(readlink -f $c); ls $c/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; done 2>&1 | head -40
python bench.py --steps 1 --warmup 1 --ddnm-steps 10 --no-cpu-baseline > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err; tail -3 gpurun_out/bench_short.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_short.json').read().strip().splitlines()[-1])
print(json.dumps(d['roofline'].get('calibration'), indent=1))
print({k:v for k,v in d['roofline'].items() if k not in ('calibration','attention')})
PY
