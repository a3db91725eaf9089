"""Times the complete_unseen_by branches of the pipeline at BASELINE sizes ('nearest' inpainting so the UNet is out of the way)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointdreamer_amd import synthetic, pipeline
import pointdreamer_amd.camera_utils as cu
dev = torch.device('cuda', 0)
V, RES, CAM_RES, A = 8, 256, 512, 1024
sh = synthetic.make_shape(30000, A)
uvs, fuv = synthetic.uv_sphere_uvs(50, 100, A, gutter=2)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = {k: T(v) for k, v in sh.items()}
cams, base_dirs, eyes, ups = cu.create_cameras(V, 1.6, CAM_RES, device=dev)
xat = dict(gb_pos=g['gb_pos'], mask=g['mask'], per_atlas_pixel_face_id=g['per_atlas_pixel_face_id'], uvs=T(uvs), mesh_tex_idx=T(fuv))
cam = dict(cams=cams, base_dirs=base_dirs, eye_positions=eyes, up_dirs=ups)
kw = dict(view_num=V, res=RES, cam_res=CAM_RES, point_validation_by_o3d=True, texture_gen_method='nearest', point_size=1, edge_point_size=1,
          crop_img=True, crop_padding=0.05, mask_ratio_thresh=0.82, edge_dilate_kernels=[21])
for mode, opt in (('unproject', None), ('neighbor', None), ('neighbor', 'ours')):
    for _ in range(2):
        out = pipeline.colorize_one_mesh(g['points'], g['colors'], g['vertices'], g['faces'], g['f_normals'], xat, cam,
                                         complete_unseen_by=mode, optimize_from=opt, return_intermediates=True, **kw)
    torch.cuda.synchronize(); t = time.time()
    out = pipeline.colorize_one_mesh(g['points'], g['colors'], g['vertices'], g['faces'], g['f_normals'], xat, cam,
                                     complete_unseen_by=mode, optimize_from=opt, return_intermediates=True, **kw)
    torch.cuda.synchronize()
    print(f"complete_unseen_by={mode} optimize_from={opt}: {1e3 * (time.time() - t):.1f} ms per shape ('nearest' inpainting); "
          f"unpainted chart texels {(~out['painted'] & g['mask'][0, :, :, 0]).sum().item()}")
