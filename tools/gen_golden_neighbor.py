"""Golden vectors for SURVEY 8(f)-2 by RUNNING THE REFERENCE's paint_invisible_areas_by_neighbors (pointdreamer/unproject.py:93-196,
with utils/mesh_utils.subdivide_with_uv) on CPU tensors.  Build container only:  python -m tools.gen_golden_neighbor
trimesh / kaolin are not installed: tools/ref_harness.py supplies restatements of trimesh.grouping.unique_rows,
trimesh.geometry.faces_to_edges and kaolin.ops.mesh.uniform_laplacian (their published behaviour).
Writes tests/golden/neighbor_{small,seam}.npz: inputs + the reference's subdivided mesh, vertex colours after the diffusion,
the atlas handed to the final nearest fill and the returned atlas."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import ref_harness as rh                      # noqa: E402
from pointdreamer_amd import synthetic                     # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def case(name, stacks, slices, A, hole, seed):
    torch.set_num_threads(1)
    _, up, _ = rh.import_reference()
    rng = np.random.default_rng(seed)
    verts, faces, lut = synthetic.uv_sphere(stacks, slices)
    uvs, fuv = synthetic.uv_sphere_uvs(stacks, slices, A, gutter=2)
    gb_pos, mask, fid = synthetic.latlong_atlas(A, stacks, slices, gutter=2, lut=lut)
    chart = mask[0, :, :, 0]
    atlas = rng.uniform(0, 1, (A, A, 3)).astype(np.float32)
    painted = chart.copy()
    r0, r1, c0, c1 = hole
    painted[r0:r1, c0:c1] = False                              # an unseen patch (no view painted it)
    painted &= rng.uniform(0, 1, (A, A)) > 0.02                # plus scattered unpainted texels
    atlas[~painted] = 0.0
    tif = np.unique(fid[0][~painted])
    tif = tif[tif > -1]
    captured = {}

    def fake_naive(img, no_need_inpaint_mask2, method='linear'):
        captured['img'] = img.detach().cpu().numpy().copy()
        captured['mask'] = no_need_inpaint_mask2.detach().cpu().numpy().copy()
        return real_naive(img, no_need_inpaint_mask2, method)
    real_naive = up.naive_inpainting
    up.naive_inpainting = fake_naive
    T = torch.from_numpy
    sv, sf, sc = up.paint_invisible_areas_by_neighbors(T(verts), T(faces), T(uvs), T(fuv), T(tif), T(atlas.copy()), T(painted.copy()),
                                                        use_atlas=False)
    out = up.paint_invisible_areas_by_neighbors(T(verts), T(faces), T(uvs), T(fuv), T(tif), T(atlas.copy()), T(painted.copy()),
                                                use_atlas=True)
    up.naive_inpainting = real_naive
    np.savez_compressed(os.path.join(OUT, name), vertices=verts, faces=faces, uvs=uvs, face_uv_idx=fuv, to_inpaint_face_id=tif,
                        atlas=atlas, painted=painted, ref_sub_vertices=sv.numpy(), ref_sub_faces=sf.numpy(),
                        ref_vert_colors=sc.numpy(), ref_atlas_before_fill=captured['img'].transpose(1, 2, 0),
                        ref_mask_before_fill=captured['mask'][0], ref_atlas=out.numpy())
    print(name, 'faces to inpaint', len(tif), 'subdivided V', len(sv), 'F', len(sf))


if __name__ == '__main__':
    assert rh.available()
    case('neighbor_small.npz', 8, 12, 64, (20, 36, 10, 30), 1)
    case('neighbor_seam.npz', 10, 16, 96, (30, 60, 80, 96), 2)       # hole touching the seam / chart edge
