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
ap.add_argument('--zeros', action='store_true', help='zero activations and weights (DVFS reference)')
ap.add_argument('--stamps', action='store_true', help='lab_stamp build: print per-phase cycle averages of the last launch')
ap.add_argument('--res', action='store_true', help='pass a residual tensor (the out_layers conv of a ResBlock)')
ap.add_argument('--apply', action='store_true', help='time the APPLY (GroupNorm + SiLU in LDS) variant of the halo kernel on 3x3 shapes')
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
    if a.zeros:
        x.zero_(); w.zero_()
    y = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev)
    fl = 2.0 * N * H * W * Cout * taps * Cin
    rt = torch.randn((N, H, W, Cout), device=dev).half() if a.res else None
    rs = P(rt) if a.res else None
    res = []
    for cfg in a.cfg:
        bk, st, wm = (int(v) for v in cfg.split('x'))
        L.pdhip_debug_set_conv_bk(bk); L.pdhip_debug_set_conv_stages(st); L.pdhip_debug_set_conv_tile(wm)
        if a.apply and taps == 9:
            tab = torch.randn((N, Cin // 8, 16), device=dev) * 0.5 + 1.0
            if a.zeros: tab.zero_()
            fn = C.CDLL(_lib.LIB_PATH).pdhip_debug_conv3x3_apply
            fn.argtypes = [C.c_void_p] * 6 + [C.c_int] * 6 + [C.c_void_p] * 2
            call = lambda st: fn(P(x), P(tab), P(w), P(b), rs, P(y), N, H, W, Cin, Cout, pad, P(zp), st)
        else:
            call = lambda st: L.pdhip_conv2d_nhwc_f16(P(x), P(w), P(b), rs, P(y), N, H, W, Cin, Cout, pad, taps, P(zp), st)
        for _ in range(3):
            call(None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            call(C.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        res.append(f"{cfg}: {fl/ms/1e9:6.0f}")
    print(f"N{N} {H}x{W} Cin{Cin} Cout{Cout} taps{taps} ({fl/1e9:7.1f} GFLOP)  " + " | ".join(res))
    if a.stamps:
        import numpy as np
        nb = min(4096, ((N * H * W + 255) // 256) * (pad // 256))
        buf = (C.c_ulonglong * (8 * nb))()
        L._handle if False else None
        fn = C.CDLL(_lib.LIB_PATH).pdhip_lab_read_stamps
        assert fn(buf, 8 * nb) == 0
        st = np.array(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
        d = np.diff(st[:, :6], axis=1)
        names = ['setup', 'mainloop', 'acc->lds', 'sync', 'readback+store']
        print('   phases (s_memtime ticks @100MHz?, avg over blocks): ' + '  '.join(f"{nm} {d[:, i].mean():9.1f}" for i, nm in enumerate(names)) + f"  total {(st[:, 5] - st[:, 0]).mean():9.1f}  | in mainloop (wave 0): hand-over waitcnt {st[:, 6].mean():9.1f}  barrier {st[:, 7].mean():9.1f}")
        tb = (C.c_ulonglong * (8 * 4096))()
        assert fn(tb, 8 * 4096) == 0
        tr = np.array(tb, dtype=np.uint64).astype(np.int64)[7 * 4096:7 * 4096 + 64].reshape(2, 32)[:, :17]
        base = tr[:, 0].min()
        print('   group start ticks, K-step it0+10 of block 8 (rows: wave 0, wave 4):')
        for r in tr: print('      ' + ' '.join(f"{v - base:5d}" for v in r))
        print('   block start spread: min %d max %d; end spread: min %d max %d' % (st[:, 0].min() - st[:, 0].min(), st[:, 0].max() - st[:, 0].min(), st[:, 5].min() - st[:, 0].min(), st[:, 5].max() - st[:, 0].min()))
L.pdhip_debug_set_conv_bk(0); L.pdhip_debug_set_conv_stages(0); L.pdhip_debug_set_conv_tile(0)
