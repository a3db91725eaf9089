"""8 shapes per step through pointdreamer_amd/shapes.py (one launch per stage for all 64 views) at BASELINE sizes: ms per shape.
Usage (GPU box): python tools/time_shapes.py [--shapes 8] [--iters 30] [--hpr 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointdreamer_amd import synthetic, shapes as shp, _lib
import pointdreamer_amd.camera_utils as cu
ap = argparse.ArgumentParser()
ap.add_argument('--shapes', type=int, default=8)
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--hpr', type=int, default=1)
a = ap.parse_args()
_lib.lib()
dev = torch.device('cuda:0')
V, RES, CAM, A = 8, 256, 512, 1024
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
sh = synthetic.make_shape(30000, A, seed=0)
g = {k: T(v) for k, v in sh.items()}
cams, base_dirs, eyes, ups = cu.create_cameras(V, 1.6, CAM, device=dev)
ci = dict(cams=cams, base_dirs=base_dirs, eye_positions=eyes, up_dirs=ups)
xa = dict(gb_pos=g['gb_pos'], mask=g['mask'], per_atlas_pixel_face_id=g['per_atlas_pixel_face_id'])
b = []
for k in range(a.shapes):
    sk = synthetic.make_shape(30000, A, seed=7000 + k)
    b.append(dict(coords=T(sk['points']), colors=T(sk['colors']), vertices=g['vertices'], faces=g['faces'], f_normals=g['f_normals'], xatlas=xa))
st = shp.stack(b)
kw = dict(texture_gen_method='nearest', point_size=1, edge_point_size=1, crop_img=True, crop_padding=0.05, mask_ratio_thresh=0.82,
          edge_dilate_kernels=[21], point_validation_by_o3d=bool(a.hpr))
for _ in range(3):
    shp.colorize_shapes(st, ci, V, RES, CAM, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.iters):
    shp.colorize_shapes(st, ci, V, RES, CAM, **kw)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
print(f"{a.shapes} shapes per step, hidden-point removal {'on' if a.hpr else 'off'}: {dt * 1e3:.3f} ms per step = {dt / a.shapes * 1e3:.4f} ms per shape")
