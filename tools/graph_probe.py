"""Probe: the 'nearest' texturing path of one shape captured into a HIP graph (torch.cuda.CUDAGraph) -- does it capture, does a replay
equal the eager result, and what do S graphs on S streams deliver per shape?  Usage (GPU box): python tools/graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointdreamer_amd import synthetic, pipeline, _lib
import pointdreamer_amd.camera_utils as cu
_lib.lib()
dev = torch.device('cuda:0')
V, RES, CAM, A = 8, 256, 512, 1024
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sh = synthetic.make_shape(30000, A, seed=0)
g = {k: T(v) for k, v in sh.items()}
cams, base_dirs, eyes, ups = cu.create_cameras(V, 1.6, CAM, device=dev)
ci = dict(cams=cams, base_dirs=base_dirs, eye_positions=eyes, up_dirs=ups)
xa = dict(gb_pos=g['gb_pos'], mask=g['mask'], per_atlas_pixel_face_id=g['per_atlas_pixel_face_id'], uvs=None, mesh_tex_idx=None)
cfg = dict(view_num=V, res=RES, cam_res=CAM, point_validation_by_o3d=True, texture_gen_method='nearest', point_size=1, edge_point_size=1,
           crop_img=True, crop_padding=0.05, mask_ratio_thresh=0.82, optimize_from=None, edge_dilate_kernels=[21],
           complete_unseen_by='unproject', inpainter=None)
one = lambda pts, col: pipeline.colorize_one_mesh(pts, col, g['vertices'], g['faces'], g['f_normals'], xa, ci, **cfg)[4]
clouds = [synthetic.make_shape(30000, A, seed=100 + i) for i in range(8)]
clouds = [(T(c['points']), T(c['colors'])) for c in clouds]
for _ in range(3):
    ref = [one(p, c).clone() for p, c in clouds]
torch.cuda.synchronize()
S = 8
slots = []
for s in range(S):
    st = torch.cuda.Stream()
    pts, col = clouds[s][0].clone(), clouds[s][1].clone()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        for _ in range(2):
            one(pts, col)
        torch.cuda.current_stream().synchronize()
        with torch.cuda.graph(gr, stream=st):
            out = one(pts, col)
    slots.append((st, gr, pts, col, out))
torch.cuda.synchronize()
print('captured', S, 'graphs')
# correctness: feed cloud (s + 1) % 8 into slot s
for s, (st, gr, pts, col, out) in enumerate(slots):
    p, c = clouds[(s + 1) % 8]
    with torch.cuda.stream(st):
        pts.copy_(p); col.copy_(c); gr.replay()
torch.cuda.synchronize()
ok = all(torch.equal(slots[s][4], ref[(s + 1) % 8]) for s in range(S))
print('replay == eager:', ok)
for ns in (1, 2, 4, 8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    iters = 50
    for _ in range(iters):
        for s in range(ns):
            st, gr, pts, col, out = slots[s]
            with torch.cuda.stream(st):
                pts.copy_(clouds[s][0]); col.copy_(clouds[s][1]); gr.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"{ns} graphs on {ns} streams: {dt * 1e3:.3f} ms per step = {dt / ns * 1e3:.3f} ms per shape")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    one(*clouds[0])
torch.cuda.synchronize(); print(f"eager, one shape at a time: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per shape")
