"""Micro-benchmark: GroupNorm-apply + skip 1x1 of a channel-changing ResBlock as ONE pass over the (virtual concat) input (k_gn_skip)
against the two launches it replaces (k_gn_apply on the two-source input is approximated by the single-tensor apply of the same
bytes, k_conv_igemm<1> on the two-source input).  Prints us per launch and HBM GB/s of the algorithmic bytes.
Usage (GPU box): python tools/bench_gnskip.py [lib.so]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
dev = 'cuda:0'
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, W, Ca, Cb) in [(32, 256, 256, 256, 256), (8, 256, 256, 256, 256), (32, 128, 128, 512, 256), (32, 128, 128, 256, 256), (8, 128, 128, 256, 256), (1, 256, 256, 256, 256)]:
    Cc = Ca + Cb
    x = torch.randn((N, H, W, Cc), device=dev).half()
    xa = x[..., :Ca].contiguous(); xb = x[..., Ca:].contiguous()
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    wp = (torch.randn((256, Cc), device=dev) / Cc ** 0.5).half(); b = torch.zeros(256, device=dev)
    stats = torch.empty((N * 64,), device=dev); ws = torch.empty((N * 64 * ((H * W + 255) // 256),), device=dev)
    y = torch.empty_like(x); sk = torch.empty((N, H, W, 256), device=dev, dtype=torch.float16)
    zp = torch.zeros((128,), dtype=torch.float16, device=dev)
    assert L.pdhip_groupnorm_nhwc_f16(P(x), P(gamma), P(beta), None, N, H, W, Cc, 1, 0, P(y), P(stats), P(ws), ws.numel(), st()) == 0
    del x
    h0 = y
    byt = N * H * W * (Cc * 2 * 2 + 256 * 2)
    for v in (0, 1):
        L.pdhip_debug_set_gn_skip_variant(v)
        t_f = timeit(lambda: L.pdhip_gn_silu_skip1x1_nhwc_f16(P(xa), P(xb), Ca, Cc, P(stats), P(gamma), P(beta), P(wp), P(b), P(h0), P(sk), N, H, W, st()))
        print(f"N{N} {H}x{W} C{Ca}+{Cb} variant {v}: one pass {t_f:8.1f} us = {byt / t_f / 1e3:6.0f} GB/s  ({byt / 1e6:.0f} MB algorithmic)", flush=True)
    del xa, xb, y, sk
    torch.cuda.empty_cache()
