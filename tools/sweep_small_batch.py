"""Small-batch UNet routing sweep (view-parallel runs batch 1-2 per rank): forward latency under the conv routing hooks.
Usage (GPU box): python tools/sweep_small_batch.py [--batches 1 2 4]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pointdreamer_amd.ddnm_inpainting as di
from pointdreamer_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument('--batches', type=int, nargs='*', default=[1, 2, 4]); a = ap.parse_args()
L = _lib.lib(); dev = torch.device('cuda:0')
sd = di.random_state_dict(dict(di.IMAGENET_256), seed=0)
rows = []
for N in a.batches:
    m = di.UNetModel(max_batch=N, device=dev, **di.IMAGENET_256); m.load_state_dict(sd)
    x = torch.randn((N, 3, 256, 256), device=dev); t = torch.full((N,), 500.0, device=dev)
    def timed(iters=10):
        for _ in range(3): m(x, t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): m(x, t)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    ws = torch.empty((16 * 384 * 128 * 128,), device=dev)
    for name, tile, fuse in (('auto', 0, 0), ('halo-everywhere', 32, 0), ('igemm-128x128', 2, 0), ('auto+fused-gn', 0, 1), ('halo-everywhere+fused-gn', 32, 1)):
        L.pdhip_debug_set_conv_tile(tile); L.pdhip_debug_set_fuse_gn(fuse)
        ms = timed()
        rows.append(dict(batch=N, routing=name, forward_ms=round(ms, 3)))
        print(json.dumps(rows[-1]), flush=True)
    L.pdhip_debug_set_conv_tile(0); L.pdhip_debug_set_fuse_gn(0)
    del m; torch.cuda.empty_cache()
os.makedirs('gpurun_out', exist_ok=True)
json.dump(rows, open('gpurun_out/small_batch_sweep.json', 'w'), indent=1)
