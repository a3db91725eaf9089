#!/bin/bash
# SQ counters of every kernel of the `nearest` workload (bench.py --workload nearest), one --pmc pass per counter group
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_nearest
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload nearest --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$@" <<'PY'
import csv, glob, collections, re, sys
want = sys.argv[1:] or ['k_raster_tiles', 'k_nbf_bits', 'k_sparse_edges']
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_nearest/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
        if m and m.group(1) in want: acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sorted(v)[len(v) // 2]) for c, v in sorted(d.items())})
PY
