#!/bin/bash
# HBM-traffic counters of the dominant kernel over a real bench run (separate --pmc passes, no other trace domains).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --ddnm-steps 2 --no-cpu-baseline --no-extras > $OUT/$tag.log 2>&1
done
ls $OUT
