cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -3
O=gpurun_out/r06/ht_ab2.txt; : > $O
for rep in 1 2; do for ht in 0 1; do
  timeout 600 python tools/time_unet.py --batches 1 2 4 --iters 20 --ht $ht --out gpurun_out/r06/unet_ht$ht.json > /dev/null 2>&1
  echo "ht=$ht rep=$rep $(python -c "
import json; r=json.load(open('gpurun_out/r06/unet_ht$ht.json')); print([(x['batch'], x['forward_ms'], x.get('ddnm_step_ms')) for x in (r['rows'] if isinstance(r, dict) else r)])")" >> $O
done; done
cat $O
