"""UNet forward latency at small and large batches (rows U1 / 8e: view-parallel runs batch 1 per rank).
Usage (GPU box): python tools/time_unet.py [--batches 1 2 4 8 32] [--iters 10] -> gpurun_out/unet_latency.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pointdreamer_amd.ddnm_inpainting as di

ap = argparse.ArgumentParser()
ap.add_argument('--batches', type=int, nargs='*', default=[1, 2, 4, 8, 32])
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--fold', type=int, default=1, help='pdhip_debug_set_fold_resample')
ap.add_argument('--fuse', type=int, default=0, help='pdhip_debug_set_fuse_gn')
ap.add_argument('--fin', type=int, default=-1, help='pdhip_debug_set_fold_finalize (largest batch with in-kernel GroupNorm statistics; -1 = default)')
ap.add_argument('--sk', type=int, default=-1, help='pdhip_debug_set_conv_sk mode (0 off, 1 auto, 2 every eligible layer; -1 = default)')
ap.add_argument('--fskip', type=int, default=-1, help='pdhip_debug_set_fuse_skip mode (0 off, 1 auto, 2 always; -1 = default)')
ap.add_argument('--gsv', type=int, default=-1, help='pdhip_debug_set_gn_skip_variant (-1 = default)')
ap.add_argument('--finc', type=int, default=-1, help='pdhip_debug_set_fold_finalize_chunks (-1 = default)')
ap.add_argument('--gn-iters', type=int, default=-1, help='pdhip_debug_set_gn_iters (-1 = default)')
ap.add_argument('--rr', type=int, default=-1, help='pdhip_debug_set_conv_rr mode (0 off, 1 auto, 2 every eligible layer; -1 = default)')
ap.add_argument('--rr-gn', type=int, default=-1, help='pdhip_debug_set_rr_gn (largest width with the in-staging GroupNorm; -1 = default)')
ap.add_argument('--ht', type=int, default=-1, help='pdhip_debug_set_conv_ht mode (0 off, 1 auto, 2 every eligible layer; -1 = default)')
ap.add_argument('--ht-slabs', type=int, default=0, help='forced K-slabs of k_conv_ht (0 = automatic)')
ap.add_argument('--out', default='gpurun_out/unet_latency.json')
ap.add_argument('--graph', type=int, default=0, help='1: also time the forward and the sampler replayed from a HIP graph (torch.cuda.CUDAGraph)')
ap.add_argument('--sampler-steps', type=int, default=10, help='also time this many DDNM steps through pdhip_ddnm_sample (0 = skip)')
a = ap.parse_args()
dev = torch.device('cuda:0')
from pointdreamer_amd import _lib
if os.environ.get('PDHIP_LAB_LIB'):                      # lab builds: PDHIP_LAB_LIB=path/to/lab.so python tools/time_unet.py
    _lib.LIB_PATH = os.path.abspath(os.environ['PDHIP_LAB_LIB'])
_lib.lib().pdhip_debug_set_fuse_gn(a.fuse); _lib.lib().pdhip_debug_set_fold_resample(a.fold)
if a.gn_iters >= 0:
    _lib.lib().pdhip_debug_set_gn_iters(a.gn_iters)
if a.fin >= 0:
    _lib.lib().pdhip_debug_set_fold_finalize(a.fin)
if a.gsv >= 0:
    _lib.lib().pdhip_debug_set_gn_skip_variant(a.gsv)
if a.finc >= 0:
    _lib.lib().pdhip_debug_set_fold_finalize_chunks(a.finc)
if a.fskip >= 0:
    _lib.lib().pdhip_debug_set_fuse_skip(a.fskip, 0)
if a.sk >= 0:
    _lib.lib().pdhip_debug_set_conv_sk(a.sk, 0, 0)
if a.rr >= 0:
    _lib.lib().pdhip_debug_set_conv_rr(a.rr, 0, 0)
if a.ht >= 0:
    _lib.lib().pdhip_debug_set_conv_ht(a.ht, a.ht_slabs)
if a.rr_gn >= 0:
    _lib.lib().pdhip_debug_set_rr_gn(a.rr_gn)
sd = di.random_state_dict(dict(di.IMAGENET_256), seed=0)
rows = []
for N in a.batches:
    m = di.UNetModel(max_batch=N, device=dev, **di.IMAGENET_256)
    m.load_state_dict(sd)
    x = torch.randn((N, 3, 256, 256), device=dev)
    t = torch.full((N,), 500.0, device=dev)
    for _ in range(3):
        m(x, t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        m(x, t)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    row = dict(batch=N, forward_ms=round(ms, 3), ms_per_image=round(ms / N, 3), tflops_effective=round(2239.7 * N / ms, 1))
    if a.sampler_steps:
        inp = di.Inpainter.__new__(di.Inpainter)
        inp.device, inp.model, inp.seed, inp.n_steps, inp._images, inp.max_batch = dev, m, 1234, a.sampler_steps, 0, N
        imgs = torch.rand((N, 3, 256, 256), device=dev); masks = (torch.rand((N, 256, 256), device=dev) > 0.5).float()
        inp.inpaint_views(imgs * masks[:, None], masks)
        torch.cuda.synchronize()
        e0.record(); inp.inpaint_views(imgs * masks[:, None], masks); e1.record(); torch.cuda.synchronize()
        row['ddnm_step_ms'] = round(e0.elapsed_time(e1) / a.sampler_steps, 3)
        if a.graph:
            st = torch.cuda.Stream()
            mi, mk = (imgs * masks[:, None]).contiguous(), masks.contiguous()
            with torch.cuda.stream(st):
                inp.inpaint_views(mi, mk, first_key=0)
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    out_g = inp.inpaint_views(mi, mk, first_key=0)
                g.replay(); st.synchronize()
                ref = inp.inpaint_views(mi, mk, first_key=0)
                st.synchronize()
                row['graph_equal'] = bool(torch.equal(out_g, ref))
                e0.record(st)
                for _ in range(3):
                    g.replay()
                e1.record(st); st.synchronize()
                row['ddnm_step_ms_graph'] = round(e0.elapsed_time(e1) / (3 * a.sampler_steps), 3)
    rows.append(row)
    print(json.dumps(row), flush=True)
    del m
    torch.cuda.empty_cache()
os.makedirs('gpurun_out', exist_ok=True)
json.dump(rows, open(a.out, 'w'), indent=1)
