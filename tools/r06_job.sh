cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "conv_ht or halo_tile" 2>&1 | tail -3
O=gpurun_out/r06/ht_ab.txt; : > $O
for rep in 1 2; do for ht in 0 1; do
  timeout 600 python tools/time_unet.py --batches 1 2 4 --iters 20 --ht $ht --out gpurun_out/r06/unet_ht$ht.json > /dev/null 2>&1
  echo "ht=$ht rep=$rep $(python -c "
import json; r=json.load(open('gpurun_out/r06/unet_ht$ht.json')); print([(x['batch'], x['forward_ms'], x.get('ddnm_step_ms')) for x in (r['rows'] if isinstance(r, dict) else r)])")" >> $O
done; done
cat $O
timeout 900 python tools/bench_ht.py --batches 1 2 4 > gpurun_out/r06/bench_ht_v7b.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06/bench_ht_v7b.txt
