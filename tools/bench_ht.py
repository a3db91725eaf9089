"""Micro-benchmark of k_conv_ht (256 x 64 halo tiles, nn_conv_ht.hip) on the UNet's 64^2 / 128^2 3x3 convs at batch 1-2 next to the engine's own
routing without it (k_conv_sk / k_conv_rr).  Usage (GPU box): python tools/bench_ht.py [--batches 1 2] [--iters 40]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
ap = argparse.ArgumentParser()
ap.add_argument('--batches', type=int, nargs='*', default=[1])
ap.add_argument('--iters', type=int, default=40)
ap.add_argument('--lib', default=None)
ap.add_argument('--shapes', type=int, nargs='*', default=None)
ap.add_argument('--tag', default='')
ap.add_argument('--slabs', type=int, nargs='*', default=[1, 0], help='K-slabs of k_conv_ht to time (0 = automatic)')
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib); os.environ['PDHIP_ALLOW_LAB_BUILD'] = '1'
L = _lib.lib()
dev = 'cuda:0'
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
zp = torch.zeros(128, dtype=torch.float16, device=dev)
ws = torch.zeros((4096 + 32 * 1024 * 1024,), dtype=torch.float32, device=dev)
SHAPES = [(128, 256, 256), (128, 512, 256), (128, 768, 256), (64, 512, 512), (64, 1024, 512), (64, 256, 512), (64, 768, 512), (32, 512, 512), (128, 512, 512), (128, 256, 512)]


def timeit(fn, iters):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for si, (H, Cin, Cout) in enumerate(SHAPES):
    if a.shapes is not None and si not in a.shapes:
        continue
    for N in a.batches:
        W = H
        pad = (Cout + 127) // 128 * 128
        nb = max(2, int(300e6 // (pad * 9 * Cin * 2)) + 1)
        x = torch.randn((N, H, W, Cin), device=dev).half()
        wps = [(torch.randn((pad, 9 * Cin), device=dev) * 0.02).half() for _ in range(nb)]
        b = torch.zeros(Cout, device=dev)
        y = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev)
        gp = torch.empty((N * (H * W // 256) * (Cout // 8) * 2,), device=dev)
        ch = C.c_int(0)
        fl = 2.0 * N * H * W * Cout * 9 * Cin
        L.pdhip_debug_set_conv_splitk(P(ws), ws.numel(), 0)
        old = L.pdhip_debug_set_conv_ht(0, 0)
        t_old = timeit(lambda i: L.pdhip_conv2d_nhwc_f16(P(x), P(wps[i % nb]), P(b), None, P(y), N, H, W, Cin, Cout, pad, 9, P(zp), S()), a.iters)
        L.pdhip_debug_set_conv_splitk(None, 0, 0)
        line = f"{a.tag}N{N} {H}x{W} Cin{Cin} Cout{Cout} {fl / 1e9:6.1f} GFLOP   engine route without k_conv_ht {t_old:6.1f} us ({fl / t_old / 1e9:5.2f} PF/s)   k_conv_ht"
        for sl in a.slabs:
            L.pdhip_debug_set_conv_ht(2, sl)
            t_new = timeit(lambda i: L.pdhip_conv_ht_f16(P(x), P(wps[i % nb]), P(b), None, 0, P(y), N, H, W, Cin, Cout, pad, P(zp), P(ws), ws.numel(), P(gp), C.byref(ch), S()), a.iters)
            line += f"  slabs {sl if sl else 'auto'}: {t_new:6.1f} us ({fl / t_new / 1e9:5.2f} PF/s)"
        L.pdhip_debug_set_conv_ht(old, 0)
        print(line, flush=True)
        del wps
