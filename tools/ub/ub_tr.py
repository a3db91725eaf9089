"""ds_read_b64_tr_b16 semantics probe (tools/ub/ub_tr.hip): which LDS elements does each lane receive?  Usage (GPU box): python tools/ub/ub_tr.py"""
import ctypes as C, os, subprocess
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'ub_tr.so')
if not os.path.exists(so):
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(here, 'ub_tr.hip'), '-o', so], check=True)
L = C.CDLL(so)
L.ub_tr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
dev = 'cuda:0'
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(slots, tag):
    s = torch.tensor(slots, dtype=torch.int32, device=dev)
    out = torch.zeros(256, dtype=torch.int16, device=dev)
    assert L.ub_tr(s.data_ptr(), out.data_ptr(), st) == 0
    torch.cuda.synchronize()
    o = out.cpu().view(64, 4).tolist()
    print(tag)
    for l in range(0, 64):
        if l < 20 or l % 16 == 0: print(f"  lane {l:2d} (8-byte slot {slots[l]:3d} = elements {4*slots[l]}..{4*slots[l]+3}) ->", o[l])
run(list(range(64)), "contiguous: lane l -> slot l")
run([100 + 7 * l for l in range(64)], "scattered: lane l -> slot 100 + 7 l")
