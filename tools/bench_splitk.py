"""Split-K tuning table: conv + reduce time (us) per (tile geometry, split factor) on the UNet's small-M layers (batch 8).
Usage (GPU box): python tools/bench_splitk.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
SHAPES = [(8, 64, 64, 512, 512, 9), (8, 32, 32, 512, 512, 9), (8, 32, 32, 1024, 512, 9), (8, 16, 16, 1024, 1024, 9), (8, 16, 16, 2048, 1024, 9),
          (8, 8, 8, 1024, 1024, 9), (8, 8, 8, 2048, 1024, 9), (8, 32, 32, 512, 1536, 1), (8, 32, 32, 512, 512, 1),
          (8, 16, 16, 1024, 3072, 1), (8, 16, 16, 1024, 1024, 1), (8, 8, 8, 1024, 3072, 1), (8, 8, 8, 1024, 1024, 1)]
import argparse
ap = argparse.ArgumentParser(); ap.add_argument('--custom', type=int, nargs='*', default=None); ap.add_argument('--geos', type=int, nargs='*', default=[0, 2, 8, 32])
a = ap.parse_args()
if a.custom:
    SHAPES = [tuple(a.custom[i:i + 6]) for i in range(0, len(a.custom), 6)]
dev = 'cuda:0'
zp = torch.zeros(128, dtype=torch.float16, device=dev)
ws = torch.empty(16 * 384 * 128 * 128, dtype=torch.float32, device=dev)
iters = 20
for (N, H, W, Cin, Cout, taps) in SHAPES:
    x = torch.randn((N, H, W, Cin), device=dev).half()
    pad = (Cout + 127) // 128 * 128
    w = (torch.randn((pad, taps * Cin), device=dev) * 0.05).half()
    b = torch.zeros(Cout, device=dev)
    y = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev)
    fl = 2.0 * N * H * W * Cout * taps * Cin
    print(f"N{N} {H}x{W} Cin{Cin} Cout{Cout} taps{taps} ({fl/1e9:6.1f} GFLOP)")
    for geo in a.geos:
        row = []
        for sp in (0, 1, 2, 3, 4, 6, 8, 12, 16):
            L.pdhip_debug_set_conv_tile(geo); L.pdhip_debug_set_conv_splitk(P(ws), ws.numel(), sp)
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                L.pdhip_conv2d_nhwc_f16(P(x), P(w), P(b), None, P(y), N, H, W, Cin, Cout, pad, taps, P(zp), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                L.pdhip_conv2d_nhwc_f16(P(x), P(w), P(b), None, P(y), N, H, W, Cin, Cout, pad, taps, P(zp), st)
            e1.record(); torch.cuda.synchronize()
            row.append(f"{'auto' if sp == 0 else sp}:{e0.elapsed_time(e1) / iters * 1e3:6.1f}")
        print(f"   geo{geo}  " + "  ".join(row))
L.pdhip_debug_set_conv_tile(0); L.pdhip_debug_set_conv_splitk(None, 0, 0)
