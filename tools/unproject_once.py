"""One shape through Uq1-Uq4 (unproject_dense, 8 views, atlas 1024) a few times: the command tools/pmc_run.py profiles for the
k_texel_visibility_v / k_view_select_blend_v counters (profiles/r05_pmc_unproject.json)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointdreamer_amd import synthetic
import pointdreamer_amd.ours_utils as ou, pointdreamer_amd.unproject as up, pointdreamer_amd.camera_utils as cu
dev = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sh = synthetic.make_shape(30000, 1024)
cams, base_dirs, eyes, ups = cu.create_cameras(8, 1.6, 512, device=dev)
hard, fidx, depth, vuv, uvc, uvs, pad, puv, pdep = ou.get_rendered_hard_mask_and_face_idx_batch(
    cams, T(sh['vertices']), T(sh['faces']), T(sh['points']), None, True, 0.05)
inp = torch.rand((8, 3, 256, 256), device=dev)
sf = torch.ones(8, device=dev)
for _ in range(5):
    up.unproject_dense(inp, T(sh['f_normals']), 256, cams, 512, base_dirs, T(sh['gb_pos']), T(sh['mask']), T(sh['per_atlas_pixel_face_id']),
                       uvc, uvs, pad, sf, depth, [21], True)
torch.cuda.synchronize()
