"""Winograd F(2x2, 3x3) go / no-go for the dominant layer shape: (65 536 pixels, 256 -> 256 channels, K = 2304) per
image, batch 8.  Not a Winograd kernel -- a measured LOWER bound of the unfused form on this engine's own kernels:
  * the 16 batched GEMMs [M/4 x 256] x [256 x 256] (2.25x fewer MACs) run as 16 launches of the engine's 1x1 conv kernel;
  * the input transform (reads X once, writes the 16 transformed planes) and the output transform (reads 16 f16 planes, writes Y) are
    priced as pure streaming passes of the same byte counts (torch copy kernels: the arithmetic of B^T d B / A^T m A is free next
    to the bytes);
against the direct halo-resident conv (k_conv3x3_halo) on the same tensor.  Usage (GPU box): python tools/bench_winograd.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import _lib
import pointdreamer_amd.ddnm_inpainting  # noqa
P = lambda t: C.c_void_p(t.data_ptr())
L = _lib.lib()
dev = 'cuda:0'
N, H, W, Ci, Co = 8, 256, 256, 256, 256
zp = torch.zeros(128, dtype=torch.float16, device=dev)
x = torch.randn((N, H, W, Ci), device=dev).half()
w9 = (torch.randn((Co, 9 * Ci), device=dev) * 0.02).half()
b = torch.zeros(Co, device=dev)
y = torch.empty((N, H, W, Co), dtype=torch.float16, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


t_direct = timed(lambda: L.pdhip_conv2d_nhwc_f16(P(x), P(w9), P(b), None, P(y), N, H, W, Ci, Co, 256, 9, P(zp), st()))
# 16 GEMMs: tiles = N * (H/2) * (W/2) = M / 4 rows each
Mt = N * (H // 2) * (W // 2)
v = torch.randn((16, Mt, Ci), device=dev).half()
m = torch.empty((16, Mt, Co), dtype=torch.float16, device=dev)
wt = (torch.randn((16, Co, Ci), device=dev) * 0.05).half()


def gemms():
    for k in range(16):
        L.pdhip_conv2d_nhwc_f16(P(v[k]), P(wt[k]), P(b), None, P(m[k]), 1, Mt // 256, 256, Ci, Co, 256, 1, P(zp), st())


t_gemm = timed(gemms)
# transforms as streaming passes of their byte counts
vin = torch.empty_like(v)
t_in = timed(lambda: (vin[:4].copy_(x.view(4, -1, Ci)[:, :Mt]), vin[4:].copy_(v[4:])))          # ~ read X + write 16 planes (approx. by copies)
t_out = timed(lambda: y.view(-1).copy_(m.view(-1)[:y.numel()])) + timed(lambda: vin.copy_(m)) * 0.5   # read 16 planes (+ write Y)
fl = 2.0 * N * H * W * Co * 9 * Ci
res = dict(shape=f"N{N} {H}x{W} {Ci}->{Co}", direct_ms=t_direct, direct_tflops=fl / t_direct / 1e9,
           winograd_gemm16_ms=t_gemm, winograd_gemm_tflops_of_its_own_macs=fl / 2.25 / t_gemm / 1e9,
           input_transform_stream_ms=t_in, output_transform_stream_ms=t_out,
           winograd_unfused_lower_bound_ms=t_gemm + t_in + t_out,
           verdict="no-go" if t_gemm + t_in + t_out > t_direct else "worth a kernel")
print(json.dumps(res, indent=1))
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/winograd_probe.json', 'w'), indent=1)
