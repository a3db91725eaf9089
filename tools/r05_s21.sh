#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do python tools/bench_sk.py --no-old --shapes 4 11 10 --tiles 0 --splits 0 --kg 8 12 --stages 2 3 4 2>&1 | grep -v amdgpu; done
python tools/bench_sk.py --no-old --tiles 0 --splits 0 --kg 8 --stages 3 4 --shape 1 128 128 512 256 9 --shape 1 128 128 768 256 9 --shape 1 128 128 256 256 1 2>&1 | grep -v amdgpu
