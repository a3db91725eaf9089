"""Row-gather LDS-DMA fill rate (tools/ub/ub_rows.hip).  Usage (GPU box): python tools/ub/ub_rows.py"""
import ctypes as C, os, subprocess
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'ub_rows.so')
if not os.path.exists(so):
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(here, 'ub_rows.hip'), '-o', so], check=True)
L = C.CDLL(so)
L.ub_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = 'cuda:0'
buf = torch.randn(640 * 1024 * 1024, device=dev)            # 2.5 GB: base (< window) + 256 rows x 18 KB + 36 x 128 B stays inside
sink = torch.zeros(16, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(stride, wgs, waves, rows, reps, shared, kstep, window):
    for _ in range(2):
        L.ub_rows(buf.data_ptr(), window, stride, wgs, waves, rows, reps, sink.data_ptr(), shared, kstep, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.ub_rows(buf.data_ptr(), window, stride, wgs, waves, rows, reps, sink.data_ptr(), shared, kstep, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return wgs * rows * 128 * kstep * reps / ms / 1e6 / wgs
print("GB/s per CU, 256 workgroups (1 per CU), 256 rows of 128 B per K-step, 36 K-steps along the rows")
for shared, tag in ((1, 'all WGs gather the SAME rows (weight tile, L2 hits)'), (0, 'each WG its own rows (activation tile)')):
    print(tag)
    for stride in (128, 512, 1024, 2048, 4608, 18432):
        row = []
        for waves in (4, 8, 16):
            row.append(f"{waves:2d}w {run(stride, 256, waves, 256, 8, shared, 36, 2000 << 20):6.1f}")
        print(f"   row stride {stride:6d} B: " + "  ".join(row))
