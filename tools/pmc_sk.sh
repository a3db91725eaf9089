#!/bin/bash
# PMC passes over a tools/bench_sk.py run of one configuration (rocprofv3 --pmc with --kernel-trace only).  Usage: pmc_sk.sh "<bench_sk args>"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sk
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_sk.py $1 > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_sk/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_conv_sk' in r['Kernel_Name'] or 'k_conv_igemm' in r['Kernel_Name']:
            import re
            m = re.search(r'(k_conv_[a-z]+)(I[^E]*(?:ELi[^E]*)*E*)', r['Kernel_Name'])
            acc[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} avg {sum(v)/len(v):16.1f}  launches {len(v)}")
PY
