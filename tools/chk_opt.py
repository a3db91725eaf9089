import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_golden
import pointdreamer_amd.camera_utils as cu
from pointdreamer_amd import optimize as popt
DEV='cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for name in ["optimize_64_3_1.npz", "optimize_128_3_0.npz", "optimize_64_100_1.npz"]:
    g = load_golden(name)
    cams = [cu.Camera(p, int(g['cam_res']), DEV) for p in g['cam_params']]
    shr = T(g['shrinked']) if g['shrinked'].size else None
    for its in (1, 2, 3) if int(g['iterations'])==3 else (100,):
        if its != int(g['iterations']): continue
        a, im = popt.optimize_color(T(g['atlas0']), T(g['inpainted']), T(g['verts']), T(g['faces']), T(g['uvs']), T(g['mesh_tex_idx']), cams, None, None, None,
                                T(g['uv_centers']), T(g['uv_scales']), float(g['padding']), T(g['scale_factors']), None, shr, iterations=its, res=1024)
        d = np.abs(a.cpu().numpy() - g['ref_atlas'])
        print(name, its, 'max', d.max(), 'frac>1e-4', (d>1e-4).mean(), 'frac>1e-5', (d>1e-5).mean(), 'n>1e-4', (d>1e-4).sum(), 'img', np.abs(im[:, :, ::16, ::16].cpu().numpy() - g['ref_images_small']).max())
        idx = np.argwhere(d > 1e-4)[:5]; print(idx.tolist(), [float(d[tuple(i)]) for i in idx])
