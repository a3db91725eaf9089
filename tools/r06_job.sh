cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python tools/pmc_run.py gpurun_out/r06/pmc_attn.json --filter k_attention_t64 --sets sq lds misc -- python $GRAFT_REPO_ROOT/tools/bench_attn.py --iters 10 > gpurun_out/r06/pmc_attn.log 2>&1; tail -3 gpurun_out/r06/pmc_attn.log; python -c "
import json; d=json.load(open('gpurun_out/r06/pmc_attn.json'))['kernels']
for k,v in d.items(): print(k, {a:b for a,b in v.items() if 'frac' in a or 'util' in a or a in ('SQ_INSTS_VALU','SQ_INSTS_MFMA','SQ_INSTS_LDS','SQ_INSTS_SALU','SQ_WAVES','launches')})
"
