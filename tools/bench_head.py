"""Micro-benchmark of the fused output head (GN -> SiLU -> conv3x3, f32-equivalent) at the UNet's size.  Usage: python tools/bench_head.py [N]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
dev = 'cuda:0'
for N in ([int(a) for a in sys.argv[1:]] or [8, 32]):
    H = W = 256; Cc = 256; Cout = 6
    x = torch.randn((N, H, W, Cc), device=dev).half()
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    w = torch.randn((Cout, Cc, 3, 3), device=dev) * 0.02; b = torch.zeros(Cout, device=dev)
    y = torch.empty((N, Cout, H, W), device=dev)
    ws = torch.empty((L.pdhip_unet_head_ws_floats(N, H, W, Cc, Cout),), device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: L.pdhip_unet_head_f32(P(x), P(gamma), P(beta), P(w), P(b), N, H, W, Cc, Cout, P(y), P(ws), ws.numel(), st)
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"N{N}: head (stats + pack + fused kernel) {us:.1f} us = {x.numel() * 2 / us / 1e3:.0f} GB/s of input")
