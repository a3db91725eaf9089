#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --ddnm-steps 3 --one-device --backend gloo --shapes-per-step 1 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err; echo rc=$?
tail -3 gpurun_out/bench_2rank.err; python - <<'PY'
import json
l=open('gpurun_out/bench_2rank.json').read().strip().splitlines()
d=json.loads(l[-1]); print(d['n_gpus'], d['value'], d['scaling'], d['extras'].get('view_parallel') if d.get('extras') else None); print(d['roofline'].get('calibration',{}).get('under_load',{}).get('samples'))
PY
