"""Generate golden vectors by RUNNING THE REFERENCE (imported from /root/reference, CPU tensors).

Run in the build container only:  python -m tools.gen_golden
Writes small .npz fixtures (inputs + the reference's outputs) into tests/golden/.  Inputs that come
from un-vendored third-party code in the reference (camera transform, mesh raster) are produced by
the build's own oracle and stored as *inputs*; everything stored as an *expected output* was
computed by the reference's own functions:
  proj_sparse_*.npz   ours_utils.get_point_validation_by_depth (:153-202), get_sparse_images (:848-882)
  nearest_*.npz       ours_utils.naive_inpainting 'nearest' (:610-643)
  nbf_*.npz           utils_2d scharr/dilate (:799-845), unproject.get_shrinked_... (:429-475)
  unproject_*.npz     unproject.unproject (:201-425), unproject.dilate_atlas (:480-504)
torch.set_num_threads(1) makes the reference's duplicate-index writes deterministic (last wins).
"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import ref_harness as rh                      # noqa: E402
from oracle import camera as ocam, project as oproj       # noqa: E402
from pointdreamer_amd import synthetic                     # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


class TorchCam:
    def __init__(self, cam):
        self.cam, self.height, self.width = cam, cam.height, cam.width

    def transform(self, pts):
        return torch.from_numpy(self.cam.transform(pts.detach().cpu().numpy()))


def scene(n_points, R, stacks, slices, V, seed):
    verts, faces, lut = synthetic.uv_sphere(stacks, slices)
    xyz, rgb = synthetic.sphere_points(n_points, seed=seed)
    cams, base_dirs, eyes, ups = ocam.create_cameras(V, 1.6, R)
    pr = oproj.project_batch(cams, verts, xyz, True, 0.05)
    hard, fid, depth = oproj.rasterize(pr['pos'], faces, R)
    return dict(verts=verts, faces=faces, lut=lut, xyz=xyz, rgb=rgb, cams=cams, base_dirs=base_dirs, eyes=eyes,
                pr=pr, hard=hard, fid=fid, depth=depth)


def gen_proj_sparse(ou, name, n_points, seed, point_size=1, edge_point_size=1, V=3, R=128, r=64):
    sc = scene(n_points, R, 12, 24, V, seed)
    pr = sc['pr']
    t = torch.from_numpy
    vis_ref, pix_ref = ou.get_point_validation_by_depth(R, t(pr['point_uvs']), t(pr['point_depths']),
                                                        t(sc['depth']), offset=0.0001)
    hard_r = oproj.downsample_masks(sc['hard'], r)
    pp = oproj.point_pixels_for_res(pr['point_uvs'], r)
    sparse, m0, m2, sf = ou.get_sparse_images(t(pp).clone(), t(sc['rgb']).clone(), vis_ref.clone(), t(hard_r).clone(),
                                              None, V, r, point_size, edge_point_size, 0.82)
    np.savez_compressed(os.path.join(OUT, name),
                        cam_params=np.stack([c.params for c in sc['cams']]), vertices=sc['verts'], faces=sc['faces'],
                        points=sc['xyz'], colors=sc['rgb'], cam_res=R, res=r,
                        point_uvs=pr['point_uvs'], point_depths=pr['point_depths'], mesh_depths=sc['depth'],
                        hard_masks_R=sc['hard'], face_idxs=sc['fid'], hard_masks_r=hard_r, point_pixels_r=pp,
                        point_size=point_size, edge_point_size=edge_point_size,
                        ref_visibility=vis_ref.numpy(), ref_point_pixels_R=pix_ref.numpy(),
                        ref_sparse=sparse.numpy(), ref_mask0=m0.numpy(), ref_mask2=m2.numpy(),
                        ref_scale_factors=sf.numpy())
    return sparse.numpy(), m2.numpy()


def gen_nearest(ou, name, sparse, mask2):
    outs = []
    for i in range(sparse.shape[0]):
        outs.append(ou.naive_inpainting(torch.from_numpy(sparse[i]), torch.from_numpy(mask2[i]), method='nearest'))
    np.savez_compressed(os.path.join(OUT, name), sparse=sparse, mask2=mask2, ref_inpainted=np.stack(outs))


def gen_unproject(ou, up, u2, name, kernels, complete, A=256, V=4, R=128, r=64, seed=3, n_charts=3):
    sc = scene(500, R, 12, 24, V, seed)
    gb_pos, mask, fid = synthetic.latlong_atlas(A, 12, 24, gutter=3, n_charts=n_charts, lut=sc['lut'])
    fn = synthetic.face_normals(sc['verts'], sc['faces'])
    rng = np.random.default_rng(seed)
    inpainted = rng.uniform(0, 1, (V, 3, r, r)).astype(np.float32)
    scale_factors = np.array([1.0, 0.8125, 1.0, 0.9][:V], np.float32)
    t = torch.from_numpy
    cams = [TorchCam(c) for c in sc['cams']]
    pr = sc['pr']
    out = up.unproject(t(inpainted), t(sc['verts']), t(fn), r, cams, R, t(sc['base_dirs']),
                       t(gb_pos), t(mask), t(fid), t(pr['uv_centers']), t(pr['uv_scales']), pr['padding'],
                       t(scale_factors), t(sc['depth']), list(kernels), '/tmp/pd_golden_dbg', complete)
    atlas, shr, vids, coords, points, painted = out
    dil = up.dilate_atlas(atlas.clone(), t(mask))
    # N1/N2 on their own (float formulation of the reference)
    vis_in = (rng.uniform(0, 1, (2, 96, 96)) > 0.55)
    ed = u2.detect_edges_in_gray_by_scharr_torch_batch(t(vis_in).unsqueeze(1).float() * 255.0)
    dl = u2.dilate_torch_batch((ed.squeeze(1) > 126.5).float() * 255.0, kernel_size=7)
    np.savez_compressed(os.path.join(OUT, name),
                        cam_params=np.stack([c.params for c in sc['cams']]), base_dirs=sc['base_dirs'],
                        f_normals=fn, gb_pos=gb_pos, mask=mask, face_id=fid, inpainted=inpainted,
                        uv_centers=pr['uv_centers'], uv_scales=pr['uv_scales'], padding=pr['padding'],
                        scale_factors=scale_factors, mesh_depths=sc['depth'], cam_res=R, res=r,
                        kernels=np.array(kernels), complete=complete,
                        ref_atlas=atlas.numpy(), ref_shrinked=shr.numpy(), ref_view_ids=vids.numpy(),
                        ref_coords=coords.numpy(), ref_points=points.numpy(), ref_painted=painted.numpy(),
                        ref_dilated=np.asarray(dil.numpy(), np.float32),
                        n1_in=vis_in, n1_ref_edges_gt125=(ed > 125).numpy(), n1_ref_edges_gt126_5=(ed > 126.5).numpy(),
                        n2_ref_dilated7=(dl > 127.5).numpy())


def main():
    torch.set_num_threads(1)
    os.makedirs(OUT, exist_ok=True)
    os.makedirs('/tmp/pd_golden_dbg', exist_ok=True)
    ou, up, u2 = rh.import_reference()
    s, m2 = gen_proj_sparse(ou, 'proj_sparse_dense.npz', 2000, seed=1)
    gen_nearest(ou, 'nearest_dense.npz', s, m2)
    s, m2 = gen_proj_sparse(ou, 'proj_sparse_rescale.npz', 300, seed=2)
    gen_nearest(ou, 'nearest_rescale.npz', s, m2)
    gen_proj_sparse(ou, 'proj_sparse_ps2.npz', 600, seed=4, point_size=2, edge_point_size=2)
    gen_proj_sparse(ou, 'proj_sparse_scale_near1.npz', 1500, seed=9)   # mask_ratio branch taken but after_res == res
    gen_unproject(ou, up, u2, 'unproject_k21.npz', [21], False)
    gen_unproject(ou, up, u2, 'unproject_k21_complete.npz', [21], True)
    gen_unproject(ou, up, u2, 'unproject_k0.npz', [0], True)
    gen_unproject(ou, up, u2, 'unproject_multi.npz', [21, 11, 7], False)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
