"""Per-stage timing of the non-NN texturing stages at BASELINE sizes (30k points, 8 views, 512/256, atlas 1024) with HIP
events, and achieved GB/s against the algorithmic byte counts of SURVEY 8(d).  Usage (GPU box): python tools/time_stages.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointdreamer_amd import _lib as _l
if os.environ.get('PDHIP_LAB_LIB'):                      # lab builds: PDHIP_LAB_LIB=path/to/lab.so python tools/time_stages.py
    _l.LIB_PATH = os.path.abspath(os.environ['PDHIP_LAB_LIB'])
from pointdreamer_amd import synthetic, hpr
import pointdreamer_amd.ours_utils as ou, pointdreamer_amd.unproject as up, pointdreamer_amd.camera_utils as cu
from pointdreamer_amd import optimize as popt
dev = 'cuda:0'
V, r, R, A, N = 8, 256, 512, 1024, 30000
sh = synthetic.make_shape(N, A)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = {k: T(v) for k, v in sh.items()}
cams, base_dirs, eyes, ups = cu.create_cameras(V, 1.6, R, device=dev)
P = int(sh['mask'].sum())
F = sh['faces'].shape[0]


def timed(fn, iters=20):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, out          # microseconds


rows = []
def add(name, us, bytes_):
    rows.append(dict(stage=name, us=round(us, 1), algorithmic_MB=round(bytes_ / 1e6, 2), GBps=round(bytes_ / us / 1e3, 1)))

us, pr = timed(lambda: ou.get_rendered_hard_mask_and_face_idx_batch(cams, g['vertices'], g['faces'], g['points'], None, True, 0.05))
hard, fidx, depth, vuv, uvc, uvs, pad, puv, pdep = pr
add('P1+P2 project + raster (8 views)', us, V * (41 * N - 5 * N + F * 36 + R * R * 9))
us, hard_r = timed(lambda: ou.resize_masks(hard, r)); add('P2b mask 512->256', us, V * (R * R + r * r))
us, pv = timed(lambda: ou.get_point_validation_and_pixels(R, puv, pdep, depth, r, 0.0001)); add('P3 depth visibility + pixel coordinates (one pass, as the pipeline calls it)', us, V * N * (17 + 16))
us, pv2 = timed(lambda: hpr.hidden_point_removal(g['points'], eyes, 100, already_valid=pv[0]), 5); add('P3b hidden-point removal (certified GJK: grid shield, split-f16 MFMA coarse level, f64 working-set level)', us, V * N * 24)
pp = pv[1]
us, sp = timed(lambda: ou.get_sparse_images(pp, g['colors'], pv2, hard_r, None, V, r, 1, 1, 0.82)); add('P4-P6 sparse views', us, V * r * r * 20)
sparse, m0, m2, sf = sp
us, inp = timed(lambda: ou.get_inpainted_images(sparse, m0, m2, None, None, V, method='nearest')); add('I0 nearest inpaint', us, V * r * r * 28)
us, _ = timed(lambda: ou.get_inpainted_images(sparse, m0, m2, None, None, V, method='linear'), 5); add("I0 linear inpaint (per-pixel Delaunay triangle, f64 integer predicates; VALU-bound)", us, V * r * r * 28)
us, vis = timed(lambda: up.texel_visibility(cams, g['gb_pos'], g['mask'], uvc, uvs, pad, depth, R)); add('Uq1+Uq2 texel visibility', us, 57 * P)
pm = g['mask'][0, :, :, 0].contiguous()
us, shr = timed(lambda: up.shrink_visibility(pm, vis, [21] * 4)); add('N1-N3 NBF shrink (4 levels as the reference)', us, 8 * A * A * V)
us, un = timed(lambda: up.unproject_dense(inp, g['f_normals'], r, cams, R, base_dirs, g['gb_pos'], g['mask'], g['per_atlas_pixel_face_id'],
                                          uvc, uvs, pad, sf, depth, [21], True)); add('Uq1-Uq4 whole unproject_dense', us, (57 + 34) * P + 8 * A * A * V)
atlas = un[0]
us, dil = timed(lambda: up.dilate_atlas(atlas, g['mask'])); add('Uq5 dilate_atlas', us, A * A * 25)
# optimize_color needs per-corner uvs: lat-long parametrisation of the sphere
from pointdreamer_amd.demo import standin_geometry
import logging
vv, ff, xd = standin_geometry(g['points'], A, dev, logging.getLogger('x'))
us, oc = timed(lambda: popt.optimize_color(dil.permute(2, 0, 1).flip(1), inp, vv, ff, xd['uvs'], xd['mesh_tex_idx'], cams, None, None, None,
                                           uvc, uvs, pad, sf, None, un[1]), 3)
add('8f-1 optimize_color (100 Adam iterations, res 1024)', us, 100 * V * 1024 * 1024 * (8 + 12 + 12))
popt._DEBUG_COUNTS = [0, 0, 0, 0]
popt.optimize_color(dil.permute(2, 0, 1).flip(1), inp, vv, ff, xd['uvs'], xd['mesh_tex_idx'], cams, None, None, None, uvc, uvs, pad, sf, None, un[1], iterations=1)
rows[-1].update(masked_pixels=popt._DEBUG_COUNTS[0], active_texels=popt._DEBUG_COUNTS[1], pixels=popt._DEBUG_COUNTS[2], texels=popt._DEBUG_COUNTS[3])
popt._DEBUG_COUNTS = None
_, st = hpr.hidden_point_removal(g['points'], eyes, 100, already_valid=pv[0], return_stats=True)
rows.append(dict(stage='P3b certificate statistics (8 views x 30k points)', **st))
tot = [r_ for r_ in rows if r_['stage'].split()[0] in ('P1+P2', 'P2b', 'P3', 'P4-P6', 'I0', 'Uq1-Uq4', 'Uq5') and 'linear' not in r_['stage']]
rows.append(dict(stage='P1-P6 + I0(nearest) + Uq1-Uq5 without P3b: sum', us=round(sum(r_['us'] for r_ in tot), 1),
                 algorithmic_MB=round(sum(r_['algorithmic_MB'] for r_ in tot), 2),
                 GBps=round(sum(r_['algorithmic_MB'] for r_ in tot) * 1e3 / sum(r_['us'] for r_ in tot), 1)))
print(json.dumps(rows, indent=1))
os.makedirs('gpurun_out', exist_ok=True)
json.dump(rows, open('gpurun_out/stage_times.json', 'w'), indent=1)
