"""GroupNorm apply timed alone vs. sandwiched between 3x3 convs (the UNet's real launch pattern): shows how much of the
in-network slowdown comes from the chip state the MFMA kernels leave behind (clocks / caches)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
dev = 'cuda:0'
N, H, W, Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 256, 256, 128
x = torch.randn((N, H, W, Cc), device=dev).half(); y = torch.empty_like(x); z = torch.empty_like(x)
gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
stats = torch.empty((N * 64,), device=dev)
ws = torch.empty((N * 64 * ((H * W + 255) // 256),), device=dev)
w = (torch.randn((128, 9 * Cc), device=dev) * 0.05).half(); b = torch.zeros(Cc, device=dev)
zp = torch.zeros(128, dtype=torch.float16, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
gn = lambda: L.pdhip_groupnorm_nhwc_f16(P(x), P(gamma), P(beta), None, N, H, W, Cc, 1, 0, P(y), P(stats), P(ws), ws.numel(), st)
conv = lambda src, dst: L.pdhip_conv2d_nhwc_f16(P(src), P(w), P(b), None, P(dst), N, H, W, Cc, Cc, 128, 9, P(zp), st)
def timed(pre, fn, post, n=20):
    tot = 0.0
    for i in range(n + 3):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); post()
        torch.cuda.synchronize()
        if i >= 3: tot += e0.elapsed_time(e1)
    return tot / n * 1e3
nop = lambda: None
print(f"N{N} 256x256 C128: gn stats+apply alone {timed(nop, gn, nop):.1f} us; after a conv {timed(lambda: conv(y, x), gn, nop):.1f} us; "
      f"between convs {timed(lambda: conv(y, x), gn, lambda: conv(y, z)):.1f} us; conv alone {timed(nop, lambda: conv(y, z), nop):.1f} us")
sc = lambda: y.copy_(x)
print(f"torch copy alone {timed(nop, sc, nop):.1f} us; after a conv {timed(lambda: conv(y, z), sc, nop):.1f} us")
