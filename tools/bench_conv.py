"""Micro-benchmark of the implicit-GEMM conv kernel on the UNet's dominant layer shapes (batch 8 views).
Usage (GPU box): python tools/bench_conv.py [--bk 0|32|64]"""
import argparse, ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
P = lambda t: C.c_void_p(t.data_ptr())
SHAPES = [  # (N, H, W, Cin, Cout, taps)
    (8, 256, 256, 256, 256, 9), (8, 256, 256, 512, 256, 9), (8, 128, 128, 256, 256, 9), (8, 64, 64, 512, 512, 9),
    (8, 64, 64, 1024, 512, 9), (8, 32, 32, 512, 512, 9), (8, 128, 128, 512, 256, 9), (8, 256, 256, 512, 256, 1),
    (8, 32, 32, 512, 1536, 1)]
ap = argparse.ArgumentParser(); ap.add_argument('--cfg', nargs='*', default=['64x2x8', '64x3x16', '64x2x16', '64x2x8']); ap.add_argument('--iters', type=int, default=20); ap.add_argument('--shapes', type=int, nargs='*', default=None)
ap.add_argument('--custom', type=int, nargs='*', default=None, help='extra shapes as N H W Cin Cout taps ...')
ap.add_argument('--lib', default=None, help='alternative libpdhip.so (lab builds)')
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
L = _lib.lib()
if a.custom:
    SHAPES = [tuple(a.custom[i:i + 6]) for i in range(0, len(a.custom), 6)]
dev = 'cuda:0'
zp = torch.zeros(128, dtype=torch.float16, device=dev)
for si, (N, H, W, Cin, Cout, taps) in enumerate(SHAPES):
    if a.shapes is not None and si not in a.shapes:
        continue
    x = torch.randn((N, H, W, Cin), device=dev).half()
    pad = (Cout + 127) // 128 * 128
    w = (torch.randn((pad, taps * Cin), device=dev) * 0.05).half()
    b = torch.zeros(Cout, device=dev)
    y = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev)
    fl = 2.0 * N * H * W * Cout * taps * Cin
    res = []
    for cfg in a.cfg:
        bk, st, wm = (int(v) for v in cfg.split('x'))
        L.pdhip_debug_set_conv_bk(bk); L.pdhip_debug_set_conv_stages(st); L.pdhip_debug_set_conv_tile(wm)
        for _ in range(3):
            L.pdhip_conv2d_nhwc_f16(P(x), P(w), P(b), None, P(y), N, H, W, Cin, Cout, pad, taps, P(zp), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            L.pdhip_conv2d_nhwc_f16(P(x), P(w), P(b), None, P(y), N, H, W, Cin, Cout, pad, taps, P(zp), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        res.append(f"{cfg}: {fl/ms/1e9:6.0f}")
    print(f"N{N} {H}x{W} Cin{Cin} Cout{Cout} taps{taps} ({fl/1e9:7.1f} GFLOP)  " + " | ".join(res))
L.pdhip_debug_set_conv_bk(0); L.pdhip_debug_set_conv_stages(0); L.pdhip_debug_set_conv_tile(0)
