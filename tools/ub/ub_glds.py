"""Per-CU fill-rate micro-benchmark (tools/ub/ub_glds.hip).  Usage (GPU box): python tools/ub/ub_glds.py"""
import ctypes as C, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'ub_glds.so')
if not os.path.exists(so):
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(here, 'ub_glds.hip'), '-o', so], check=True)
L = C.CDLL(so)
L.ub_fill.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
dev = 'cuda:0'
buf = torch.randn(64 * 1024 * 1024, device=dev)             # 256 MB
sink = torch.zeros(16, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(mode, window, wgs, waves, bpw, reps, same):
    for _ in range(2):
        L.ub_fill(mode, buf.data_ptr(), window, wgs, waves, bpw, reps, sink.data_ptr(), same, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.ub_fill(mode, buf.data_ptr(), window, wgs, waves, bpw, reps, sink.data_ptr(), same, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return wgs * bpw * reps / ms / 1e6                          # GB/s total
names = ['glds->LDS', 'load->VGPR', 'load->VGPR->ds_write']
for same, window, tag in ((1, 1 << 20, 'every WG reads the SAME 1 MB (L2 hits)'), (0, 16 << 20, 'WGs read disjoint slices of a 16 MB window (L2-resident)'),
                          (0, 256 << 20, 'disjoint slices of 256 MB (HBM / MALL)')):
    print(tag)
    for wgs in (256, 64):
        for mode in (0, 1, 2):
            row = []
            for waves in (1, 2, 4, 8, 16):
                bpw = 512 * 1024 if same else (1 << 20 if window > (16 << 20) else 64 * 1024)
                if same: bpw = 1 << 20
                reps = 8 if bpw >= (1 << 20) else 64
                gbs = run(mode, window, wgs, waves, bpw, reps, same)
                row.append(f"{waves:2d}w {gbs / wgs:6.1f}")
            print(f"  {wgs:3d} WGs {names[mode]:22s} GB/s per CU: " + "  ".join(row))
