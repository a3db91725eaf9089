#!/bin/bash
# SQ stall breakdown of the conv kernel (micro-benchmark shapes given as $1.., default "0 3")
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv2
rm -rf $OUT; mkdir -p $OUT
SHAPES=${@:-0 3}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --cfg 64x2x8 --iters 3 --shapes $SHAPES > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_conv_igemm' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:48], r.get('Grid_Size', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
