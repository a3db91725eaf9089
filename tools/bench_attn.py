"""T >= 128 attention kernel (k_attention_t64) in isolation: time per launch and effective TFLOP/s for the UNet's shapes, with
2 / 3 LDS chunk buffers and V read by LDS transpose read (tr) or from a transposed workspace (vt, + the k_transpose_v launch).  Usage (GPU box): python tools/bench_attn.py [--iters 50]"""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pointdreamer_amd.ddnm_inpainting as di
from pointdreamer_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=50)
a = ap.parse_args()
L = _lib.lib()
dev = torch.device('cuda:0')
ptr = lambda t: t.data_ptr()
stream = lambda: torch.cuda.current_stream().cuda_stream
for (N, T, C) in [(1, 1024, 512), (2, 1024, 512), (8, 1024, 512), (32, 1024, 512), (1, 256, 1024), (8, 256, 1024), (32, 256, 1024)]:
    D = 64
    g = torch.Generator().manual_seed(T + C)
    qkv = (torch.randn((N, T, 3 * C), generator=g) * 1.5).half().to(dev)
    out = torch.empty((N, T, C), dtype=torch.float16, device=dev)
    vt = torch.empty((N, T, C), dtype=torch.float16, device=dev)
    line = f'N{N} T{T} C{C}: {4.0 * N * T * T * C / 1e9:7.2f} GFLOP '
    ref = None
    for nbuf, vtf, qt in ((3, 1, 0), (2, 0, 1), (3, 0, 1), (2, 0, 2), (3, 0, 2)):
        L.pdhip_debug_set_attn(nbuf, vtf, qt)
        for _ in range(5):
            assert L.pdhip_attention_f16(ptr(qkv), ptr(out), N, T, C, D, ptr(vt), stream()) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            L.pdhip_attention_f16(ptr(qkv), ptr(out), N, T, C, D, ptr(vt), stream())
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        if ref is None: ref = out.clone()
        same = bool((out == ref).all())
        line += f' | {"vt" if vtf else "tr"}{nbuf}q{qt}: {us:6.1f} us {4.0 * N * T * T * C / us / 1e6:5.1f} TF {"=" if same else "DIFF"}'
    L.pdhip_debug_set_attn(0, 0, 0)
    print(line, flush=True)
