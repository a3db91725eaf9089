#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python - <<'PY'
import torch, json, sys
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.calibrate(torch.device('cuda:0'), seconds=6.0), indent=1))
PY
