"""Does a GroupNorm pass run faster when its input was just written by a conv of a small image group (tensor still in the
256 MB Infinity Cache)?  conv3x3 (halo kernel) -> GroupNorm+SiLU on its output, 256^2 x 256 channels, N images per launch;
reports the GroupNorm time per image and the achieved GB/s (read + write of the tensor)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
L = _lib.lib(); dev = 'cuda:0'
P = lambda t: C.c_void_p(t.data_ptr())
H = W = 256; Cc = 256
zp = torch.zeros(128, dtype=torch.float16, device=dev)
w = (torch.randn((Cc, 9 * Cc), device=dev) * 0.02).half(); b = torch.zeros(Cc, device=dev)
gam = torch.ones(Cc, device=dev); bet = torch.zeros(Cc, device=dev)
for N in (1, 2, 4, 8, 16, 32):
    x = torch.randn((N, H, W, Cc), device=dev).half()
    y = torch.empty_like(x); h = torch.empty_like(x)
    st = torch.empty((N * 64,), device=dev); ws = torch.empty((N * 64 * 256,), device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def conv(): L.pdhip_conv2d_nhwc_f16(P(x), P(w), P(b), None, P(y), N, H, W, Cc, Cc, Cc, 9, P(zp), s)
    def gn(): L.pdhip_groupnorm_nhwc_f16(P(y), P(gam), P(bet), None, N, H, W, Cc, 1, 0, P(h), P(st), P(ws), ws.numel(), s)
    for _ in range(3): conv(); gn()
    torch.cuda.synchronize()
    tg = tc = 0.0
    for _ in range(10):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); conv(); e1.record(); gn(); e2.record(); torch.cuda.synchronize()
        tc += e0.elapsed_time(e1); tg += e1.elapsed_time(e2)
    tc /= 10; tg /= 10
    gb = N * H * W * Cc * 2 * 3 / 1e9          # stats read + apply read + write
    print(f"N {N:2d}: conv {tc*1e3/N:7.1f} us/img ({2*N*H*W*Cc*9*Cc/tc/1e9:6.0f} TF)   GN stats+apply {tg*1e3/N:6.1f} us/img  {gb/tg*1e3:7.0f} GB/s  (tensor {N*H*W*Cc*2/1e6:.0f} MB)")
