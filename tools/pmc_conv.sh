#!/bin/bash
# PMC passes for the conv kernel micro-benchmark (separate passes; counters only with --kernel-trace).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv
mkdir -p $OUT
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --cfg 64x2 --iters 3 --shapes 0 1 > $OUT/$tag.log 2>&1
done
ls -R $OUT | head -40
