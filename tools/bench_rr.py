"""Micro-benchmark of the row-resident conv (nn_conv_rr.hip) on the UNet's 8^2 / 16^2 / 32^2 ResBlock convs, next to what it replaces:
k_gn_apply (in-kernel statistics) + k_conv_sk.  Weights rotate over enough copies to exceed the 256 MB Infinity Cache (a DDNM step reads each
layer's weights once: 1.1 GB per forward), unless --fixed.  Usage (GPU box):
  python tools/bench_rr.py [--shapes 0 1 ..] [--variants 0 ..] [--slabs 0 1 2 4 8] [--batches 1 2 4 8] [--no-gn] [--lib path]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
SHAPES = [  # (H, Cin, Cout, Cs) -- Cs: channels of the block input when the ResBlock's skip 1x1 rides along (decoder conv2), else 0
    (8, 1024, 1024, 0), (8, 2048, 1024, 0), (8, 1024, 1024, 2048),
    (16, 1024, 1024, 0), (16, 2048, 1024, 0), (16, 512, 1024, 0), (16, 1024, 1024, 1536),
    (32, 512, 512, 0), (32, 1024, 512, 0), (32, 512, 512, 768), (32, 256, 512, 0),
    (64, 512, 512, 0), (64, 1024, 512, 0), (64, 256, 512, 0), (64, 512, 512, 768), (128, 256, 256, 0), (128, 512, 256, 0), (128, 256, 256, 512)]
ap = argparse.ArgumentParser()
ap.add_argument('--shapes', type=int, nargs='*', default=None)
ap.add_argument('--variants', type=int, nargs='*', default=[0])
ap.add_argument('--slabs', type=int, nargs='*', default=[0])
ap.add_argument('--batches', type=int, nargs='*', default=[1])
ap.add_argument('--iters', type=int, default=40)
ap.add_argument('--fixed', action='store_true')
ap.add_argument('--no-gn', action='store_true', help='raw input (gn mode 0): the conv alone')
ap.add_argument('--no-old', action='store_true')
ap.add_argument('--taps', type=int, default=9, help='1: the 1x1 convs (qkv / proj shapes via --shape)')
ap.add_argument('--gn-mode', type=int, default=2, help='1 = GroupNorm without SiLU (attention norm), 2 = with SiLU')
ap.add_argument('--shape', type=int, nargs=4, action='append', default=None, metavar=('H', 'CIN', 'COUT', 'CS'))
ap.add_argument('--lib', default=None)
ap.add_argument('--stamps', action='store_true', help='lab stamp build (tools/lab_unit.sh NAME nn_conv_rr -DPD_LAB_RR_STAMP): per-phase cycles of wave 0, median over the workgroups')
a = ap.parse_args()
if a.shape:
    SHAPES = [tuple(x) for x in a.shape]; a.shapes = None
TAPS = a.taps
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
    os.environ['PDHIP_ALLOW_LAB_BUILD'] = '1'
L = _lib.lib()
dev = 'cuda:0'
zp = torch.zeros(128, dtype=torch.float16, device=dev)
ws = torch.zeros((4096 + 32 * 1024 * 1024,), dtype=torch.float32, device=dev)
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


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


for si, (H, Cin, Cout, Cs) in enumerate(SHAPES):
    if a.shapes is not None and si not in a.shapes:
        continue
    for N in a.batches:
        W = H
        K = TAPS * Cin + Cs
        pad = (Cout + 127) // 128 * 128
        wbytes = Cout * K * 2
        nb = 1 if a.fixed else max(2, int(400e6 // wbytes) + 1)
        x = (torch.randn((N, H, W, Cin), device=dev) * 1.2).half()
        xs = (torch.randn((N, H, W, Cs), device=dev)).half() if Cs else None
        gamma = torch.ones(Cin, device=dev); beta = torch.zeros(Cin, device=dev)
        film = (0.1 * torch.randn((N, 2 * Cin), device=dev))
        part = torch.empty((N * (Cin // 8) * 2,), device=dev)
        assert L.pdhip_gn_octet_partials_f16(P(x), N, H * W, Cin, 1, P(part), S()) == 0
        wps = [(torch.randn((pad, K), device=dev) * 0.02).half() for _ in range(nb)]
        wfs = []
        for wp in wps:
            wf = torch.empty((L.pdhip_conv_rr_weight_halfs(Cin, TAPS, Cs, Cout),), dtype=torch.float16, device=dev)
            assert L.pdhip_conv_rr_pack_f16(P(wp), Cin, TAPS, Cs, Cout, P(wf), S()) == 0, L.pdhip_last_error()
            wfs.append(wf)
        b = torch.zeros(Cout, device=dev)
        y = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev)
        h = torch.empty_like(x)
        gp = torch.empty((N * 16 * (Cout // 8) * 2,), device=dev)
        ch = C.c_int(0)
        fl = 2.0 * N * H * W * Cout * K
        gn = 0 if a.no_gn else a.gn_mode
        line = f"N{N} {H}x{W} Cin{Cin} Cout{Cout} skip{Cs}  {fl/1e9:6.1f} GFLOP  weights {wbytes/1e6:5.1f} MB x{nb}"
        if not a.no_old and Cs == 0:
            L.pdhip_debug_set_conv_splitk(P(ws), ws.numel(), 0)
            wold = [w_[:, :TAPS * Cin].contiguous() for w_ in wps]

            def old(i):
                if gn:
                    L.pdhip_gn_apply_parts_f16(P(x), None, Cin, Cin, P(part), 1, None, 0, P(gamma), P(beta), P(film) if gn == 2 else None, 2 * Cin, N, H, W, 1 if gn == 2 else 0, P(h), S())
                rc = L.pdhip_conv2d_nhwc_f16(P(h if gn else x), P(wold[i % nb]), P(b), None, P(y), N, H, W, Cin, Cout, pad, TAPS, P(zp), S())
                assert rc == 0, L.pdhip_last_error()
            prev = L.pdhip_debug_set_conv_rr(0, 0, 0)
            t_old = timeit(old, a.iters)
            L.pdhip_debug_set_conv_rr(prev, 0, 0)
            L.pdhip_debug_set_conv_splitk(None, 0, 0)
            line += f"   gn_apply + conv_sk {t_old:6.1f} us"
            del wold
        print(line)
        for v in a.variants:
            row = []
            for sl in a.slabs:
                L.pdhip_debug_set_conv_rr(2, v, sl)

                def new(i):
                    rc = L.pdhip_conv_rr_f16(P(x), None, Cin, Cin, gn, P(gamma), P(beta), P(film) if gn == 2 else None, 2 * Cin, P(part), 1, None, 0, P(xs), None, Cs, Cs, TAPS,
                                             P(wfs[i % nb]), P(b), None, 0, P(y), N, H, W, Cout, P(ws), ws.numel(), P(gp), C.byref(ch), S())
                    return rc
                if new(0) != 0:
                    row.append(f"s{sl}: n/a"); continue
                t = timeit(new, a.iters)
                row.append(f"s{sl}:{t:6.1f} us ({wbytes / t / 1e6:5.2f} TB/s, {fl / t / 1e9:5.2f} PF/s)")
                if a.stamps:
                    import numpy as np
                    buf = (C.c_ulonglong * 4096)()
                    fn = C.CDLL(_lib.LIB_PATH).pdhip_lab_rr_read_stamps
                    assert fn(buf, 4096) == 0
                    st = np.array(buf, dtype=np.uint64).astype(np.int64).reshape(256, 16)
                    st = st[st[:, 15] > 0]
                    names = ['-', 'prologue + first loads issued', 'statistics', 'sync + act wait + raw store', 'transform + next act issued + sync', 'mfma + next weights issued', 'exchange', 'publish + ticket', 'combine', 'epilogue']
                    fin = st[st[:, 9] > 0]
                    d = [f"{names[k]} {np.median(fin[:, k]):.0f}" for k in range(1, 10)] if len(fin) else []
                    row.append("\n        [cycles per phase summed over the units, median over the finishing workgroups: " + ", ".join(d) +
                               f" | sum {np.median(fin[:, 1:10].sum(1)) if len(fin) else -1:.0f}]")
            print(f"   rr variant {v}: " + "  ".join(row))
        L.pdhip_debug_set_conv_rr(1, 0, 0)
        del wps, wfs
