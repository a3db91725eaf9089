"""Micro-benchmark of the small-M conv kernel (nn_conv_sk.hip) over tile shape x split factor x LDS stages, next to the previous
route (k_conv_igemm split-K + k_splitk_reduce).  Weights rotate over enough buffers to exceed the 256 MB Infinity Cache unless
--fixed (then the weight stream is cache-resident: the difference is what HBM costs).  Usage (GPU box):
  python tools/bench_sk.py [--shapes 0 1 ..] [--tiles 1 2 3 4] [--splits 0 1 2 4 8 16 32] [--stages 0] [--fixed]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
P = lambda t: C.c_void_p(t.data_ptr())
SHAPES = [  # (N, H, W, Cin, Cout, taps)
    (1, 8, 8, 1024, 1024, 9), (1, 16, 16, 1024, 1024, 9), (1, 32, 32, 512, 512, 9), (1, 64, 64, 512, 512, 9), (1, 128, 128, 256, 256, 9),
    (1, 8, 8, 1024, 3072, 1), (1, 8, 8, 1024, 1024, 1), (1, 32, 32, 512, 1536, 1), (1, 16, 16, 2048, 1024, 1),
    (8, 8, 8, 1024, 1024, 9), (8, 16, 16, 1024, 1024, 9), (8, 32, 32, 512, 512, 9),
    (32, 8, 8, 1024, 1024, 9), (32, 16, 16, 1024, 1024, 9), (8, 8, 8, 2048, 1024, 9), (8, 16, 16, 2048, 1024, 9),
    (8, 256, 256, 512, 256, 1), (8, 128, 128, 512, 256, 1), (32, 32, 32, 512, 1536, 1)]
ap = argparse.ArgumentParser()
ap.add_argument('--shapes', type=int, nargs='*', default=None)
ap.add_argument('--shape', type=int, nargs=6, action='append', default=None, metavar=('N', 'H', 'W', 'CIN', 'COUT', 'TAPS'), help='an extra layer (repeatable); runs instead of the table')
ap.add_argument('--no-old', action='store_true', help='skip the igemm + reduce comparison')
ap.add_argument('--tiles', type=int, nargs='*', default=[1, 2, 3, 4])
ap.add_argument('--splits', type=int, nargs='*', default=[0, 1, 2, 4, 8, 16, 32])
ap.add_argument('--stages', type=int, nargs='*', default=[0])
ap.add_argument('--kg', type=int, nargs='*', default=[0])
ap.add_argument('--order', type=int, default=0, help='tile order hook: 0 auto, 1 pixel tiles fastest, 2 n-tiles fastest')
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--fixed', action='store_true')
ap.add_argument('--stamps', action='store_true', help='lab stamp build: print the per-wave loop time split of the last launch')
ap.add_argument('--lib', default=None, help='alternative libpdhip.so (tools/lab_unit.sh NAME nn_conv_sk -D... builds)')
a = ap.parse_args()
if a.shape:
    SHAPES = [tuple(x) for x in a.shape]; a.shapes = None
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
L = _lib.lib()
L.pdhip_debug_set_conv_sk_order(a.order)
dev = 'cuda:0'
zp = torch.zeros(128, dtype=torch.float16, device=dev)
ws = torch.zeros((4096 + 64 * 1024 * 1024,), dtype=torch.float32, device=dev)
for si, (N, H, W, Cin, Cout, taps) in enumerate(SHAPES):
    if a.shapes is not None and si not in a.shapes:
        continue
    pad = (Cout + 127) // 128 * 128
    wbytes = pad * taps * Cin * 2
    nb = 1 if a.fixed else max(2, int(400e6 // wbytes) + 1)
    x = torch.randn((N, H, W, Cin), device=dev).half()
    wts = [(torch.randn((pad, taps * Cin), device=dev) * 0.05).half() for _ in range(nb)]
    b = torch.zeros(Cout, device=dev)
    y = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev)
    fl = 2.0 * N * H * W * Cout * taps * Cin

    def timed():
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(3):
            L.pdhip_conv2d_nhwc_f16(P(x), P(wts[i % nb]), P(b), None, P(y), N, H, W, Cin, Cout, pad, taps, P(zp), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.iters):
            rc = L.pdhip_conv2d_nhwc_f16(P(x), P(wts[i % nb]), P(b), None, P(y), N, H, W, Cin, Cout, pad, taps, P(zp), s)
            assert rc == 0, L.pdhip_last_error()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3
    L.pdhip_debug_set_conv_splitk(P(ws), ws.numel(), 0)
    old = L.pdhip_debug_set_conv_sk(0, 0, 0)
    t_old = 0.0 if a.no_old else timed()
    print(f"N{N} {H}x{W} Cin{Cin} Cout{Cout} taps{taps}  {fl/1e9:6.1f} GFLOP  weights {wbytes/1e6:5.1f} MB x{nb}   igemm+reduce {t_old:6.1f} us")
    for st, kg in [(s_, k_) for k_ in a.kg for s_ in a.stages]:
        L.pdhip_debug_set_conv_sk_stages(st); L.pdhip_debug_set_conv_sk_kgroups(kg)
        for tile in a.tiles:
            row = []
            for sp in a.splits:
                if (tile in (1, 2) and (H * W) % 128 != 0 and H * W != 64):
                    continue
                L.pdhip_debug_set_conv_sk(2 if tile else 1, tile, sp)
                row.append(f"s{sp}:{timed():6.1f}")
                if a.stamps:
                    import numpy as np
                    buf = (C.c_ulonglong * 1024)()
                    fn = C.CDLL(_lib.LIB_PATH).pdhip_lab_sk_read_stamps
                    assert fn(buf, 1024) == 0
                    st_ = np.array(buf, dtype=np.uint64).astype(np.int64).reshape(64, 16)[:8]
                    for r in st_:
                        n_ = max(int(r[5]), 1)
                        row.append(f"\n        [steps {n_}: per step wait {r[0]/n_:6.0f} barrier {r[1]/n_:6.0f} issue {r[2]/n_:6.0f} compute {r[3]/n_:6.0f} | loop {r[4]/n_:6.0f} cyc"
                                   f" | compute phase from its first ds_read: reads issued {r[8]/n_:5.0f}, first group may start {r[9]/n_:5.0f}, k-half 1 may start {r[10]/n_:5.0f}, last wait passed {r[11]/n_:5.0f}]")
            if row:
                print(f"   kg {kg} stages {st} tile {tile}: " + "  ".join(row))
    L.pdhip_debug_set_conv_sk_stages(0); L.pdhip_debug_set_conv_sk_kgroups(0)
    L.pdhip_debug_set_conv_sk(old, 0, 0)
    L.pdhip_debug_set_conv_splitk(None, 0, 0)
    del wts
