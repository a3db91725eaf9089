#!/bin/bash
# Same-box A/B of the headline: the round-4 tree (git archive dce086c -> build/r4tree, library built from its own sources; untracked,
# travels with the snapshot) against the working tree, bench.py defaults minus the side measurements, alternating R times.
#   mkdir -p build/r4tree && git archive dce086c bench.py pointdreamer_amd oracle configs include profiles/r04_pmc_conv.json | tar -x -C build/r4tree
#   make -C build/r4tree/pointdreamer_amd/csrc -j6
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out/r05_ab_r4_r5.txt; : > $O
R=${1:-3}
for i in $(seq 1 $R); do
  for arm in r4 r5; do
    if [ $arm = r4 ]; then d=build/r4tree; else d=.; fi
    line=$(cd $d && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
    echo "$arm run $i: $(python - "$line" <<'PY'
import json, sys
j = json.loads(sys.argv[1]); r = j['roofline']
print(f"value {j['value']:.1f} {j['unit']}  ms_per_step {j['ms_per_step']:.1f}  dominant kernel {r['achieved']:.1f} {r['unit']} frac {r['frac']:.4f}  gemm calib {r.get('calibration', {}).get('gemm_f16_random_tflops')}")
PY
)" | tee -a $O
  done
done
