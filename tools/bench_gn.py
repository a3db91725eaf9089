"""Micro-benchmark of GroupNorm(+FiLM)+SiLU (stats + apply) on the UNet's large activations; run it under
rocprofv3 --kernel-trace --stats to get the per-kernel split.  Prints effective GB/s of stats+apply together."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
dev = 'cuda:0'
for (N, H, W, Cc, film) in [(8, 256, 256, 256, 0), (8, 256, 256, 256, 1), (8, 256, 256, 512, 0), (8, 128, 128, 256, 1), (8, 128, 128, 512, 0), (8, 64, 64, 512, 1), (8, 64, 64, 1024, 0)]:
    x = torch.randn((N, H, W, Cc), device=dev).half()
    y = torch.empty_like(x)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    fl = torch.randn((N, 2 * Cc), device=dev) * 0.1 if film else None
    stats = torch.empty((N * 64,), device=dev)
    ws = torch.empty((N * 64 * ((H * W + 255) // 256),), device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        rc = L.pdhip_groupnorm_nhwc_f16(P(x), P(gamma), P(beta), P(fl), N, H, W, Cc, 1, 0, P(y), P(stats), P(ws), ws.numel(), st)
        assert rc == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"N{N} {H}x{W} C{Cc} film{film}: stats+apply {us:7.1f} us  ({x.numel() * 2 * 3 / us / 1e3:6.0f} GB/s over 3 passes)")
