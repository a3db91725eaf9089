#!/bin/bash
# PMC passes over tools/bench_sk.py runs of single layers (rocprofv3 --pmc with --kernel-trace only -- never with a trace domain).
# Usage (GPU box): tools/pmc_sk.sh OUT.json "<bench_sk args of layer 1>" "<... layer 2>" ...
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_sk"
rm -rf "$OUT"; mkdir -p "$OUT"
JSON="$1"; shift
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE"
      "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVES"
      "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum")
li=0
for layer in "$@"; do
  li=$((li+1)); si=0
  for set in "${SETS[@]}"; do
    si=$((si+1))
    rocprofv3 --kernel-trace --pmc $set -d "$OUT/l${li}_s${si}" -o pmc --output-format csv -- python "$GRAFT_REPO_ROOT/tools/bench_sk.py" $layer > "$OUT/l${li}_s${si}.log" 2>&1 || echo "pass l$li s$si failed (see $OUT/l${li}_s${si}.log)"
  done
  echo "$layer" > "$OUT/l${li}.args"
done
cd "$GRAFT_REPO_ROOT"
python - "$JSON" <<'PY'
import csv, glob, collections, json, sys, re, os
res = {}
for af in sorted(glob.glob('gpurun_out/pmc_sk/l*.args')):
    li = os.path.basename(af)[:-5]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'gpurun_out/pmc_sk/{li}_s*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_conv_sk' in r['Kernel_Name']:
                acc[r['Kernel_Name'].split('(')[0][-70:]][r['Counter_Name']].append(float(r['Counter_Value']))
    layer = {}
    for k, d in acc.items():
        c = {n: sum(v) / len(v) for n, v in d.items()}
        c['launches'] = len(next(iter(d.values())))
        # derived, per launch: SQ_* cycle counters are quad-cycles summed over waves; MFMA busy is cycles
        if 'SQ_WAVE_CYCLES' in c:
            for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM'):
                if n in c: c['frac_of_wave_cycles.' + n] = round(c[n] / c['SQ_WAVE_CYCLES'], 4)
        if 'SQ_LDS_IDX_ACTIVE' in c and 'SQ_LDS_BANK_CONFLICT' in c:
            c['lds_conflict_frac_of_active'] = round(c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), 4)
        if 'SQ_LDS_IDX_ACTIVE' in c and 'SQ_BUSY_CU_CYCLES' in c:
            c['lds_active_frac_of_cu_busy'] = round(c['SQ_LDS_IDX_ACTIVE'] / max(c['SQ_BUSY_CU_CYCLES'], 1), 4)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
            c['mfma_util'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 256 * 4), 4)
        layer[k] = c
    res[open(af).read().strip()] = layer
json.dump(res, open(sys.argv[1], 'w'), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
