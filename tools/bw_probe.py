"""Calibration of the box: HBM streaming rates of torch copy / silu / reduce / fill kernels on UNet-sized f16 tensors
(GB/s, read + write bytes) and the hipBLASLt f16 GEMM rate.  Usage (GPU box): python tools/bw_probe.py"""
import torch, sys
dev = 'cuda:0'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (134, 537, 1074):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device=dev, dtype=torch.float16); y = torch.empty_like(x)
    s = t(lambda: y.copy_(x)); print(f'copy  {mb} MB: {2 * n * 2 / s / 1e9:.0f} GB/s')
    s = t(lambda: torch.nn.functional.silu(x, inplace=False)); print(f'silu  {mb} MB: {2 * n * 2 / s / 1e9:.0f} GB/s')
    s = t(lambda: x.sum()); print(f'read  {mb} MB: {n * 2 / s / 1e9:.0f} GB/s')
    s = t(lambda: y.zero_()); print(f'write {mb} MB: {n * 2 / s / 1e9:.0f} GB/s')
# dense f16 GEMM through hipBLASLt (torch.matmul): the practical MFMA ceiling of this box under its power limit
for n in (4096, 8192, 16384):
    a = torch.randn(n, n, device=dev, dtype=torch.float16); b = torch.randn(n, n, device=dev, dtype=torch.float16)
    s = t(lambda: a @ b, 10); print(f'gemm f16 {n}^3 (randn): {2 * n ** 3 / s / 1e12:.0f} TFLOP/s')
    a.zero_(); b.zero_()
    s = t(lambda: a @ b, 10); print(f'gemm f16 {n}^3 (zeros): {2 * n ** 3 / s / 1e12:.0f} TFLOP/s')
