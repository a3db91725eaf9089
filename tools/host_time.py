"""Host-side enqueue time of one `nearest` shape (no synchronisation inside the loop) against the synchronised wall time, and a
cProfile of the enqueue loop: what the host spends per launch.  Usage (GPU box): python tools/host_time.py"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointdreamer_amd import synthetic, pipeline
import pointdreamer_amd.camera_utils as cu
dev = 'cuda:0'
sh = synthetic.make_shape(30000, 1024)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cams, base_dirs, eyes, ups = cu.create_cameras(8, 1.6, 512, device=dev)
ci = dict(cams=cams, base_dirs=base_dirs, eye_positions=eyes, up_dirs=ups)
xat = dict(gb_pos=T(sh['gb_pos']), mask=T(sh['mask']), per_atlas_pixel_face_id=T(sh['per_atlas_pixel_face_id']))
args = (T(sh['points']), T(sh['colors']), T(sh['vertices']), T(sh['faces']), T(sh['f_normals']), xat, ci, 8, 256, 512)
kw = dict(texture_gen_method='nearest', complete_unseen_by='unproject', optimize_from=None, point_validation_by_o3d=True)
for _ in range(5): pipeline.colorize_one_mesh(*args, **kw)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n): pipeline.colorize_one_mesh(*args, **kw)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / n:.3f} ms per shape, synchronised {1e3 * (t2 - t0) / n:.3f} ms per shape")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): pipeline.colorize_one_mesh(*args, **kw)
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(22); print(st.getvalue()[:5000])
