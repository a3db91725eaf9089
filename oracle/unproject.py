"""Rows Uq1-Uq4: NBF unprojection into the UV atlas.  (oracle -- test infrastructure)

Follows /root/reference/pointdreamer/unproject.py:201-425.  Arithmetic contract shared with the
HIP kernels (float32, one rounding per op, no FMA):
    uv_ns = ((xy - center)/scale) * (1-2pad) + 0.5                    (depth test, Uq1/Uq2)
    uv    = (((xy - center)/scale) * s_v) * (1-2pad) + 0.5            (colour lookup, Uq4)
    sim_v = ((n0*d0 + n1*d1) + n2*d2)                                 (Uq3)
    w_v   = f32(exp(f64(sim_v - max_v sim))) / sum_v(...)  summed in view order; first max wins.
"""
import numpy as np
from .project import point_validation_by_depth
from .nbf import shrink_visibility

F32 = np.float32


def softmax_rows_f32(sim):
    sim = np.asarray(sim, F32)
    m = sim.max(1, keepdims=True)
    e = np.exp((sim - m).astype(np.float64)).astype(F32)
    s = np.zeros((sim.shape[0],), F32)
    for v in range(sim.shape[1]):
        s = s + e[:, v]
    return e / s[:, None]


def unproject(inpainted_images, f_normals, view_img_res, cams, cam_res, base_dirs, gb_pos, mask,
              per_atlas_pixel_face_id, uv_centers, uv_scales, padding, inpaint_scale_factors,
              mesh_normalized_depths, edge_dilate_kernels, complete_unseen_by_projection=False):
    """unproject.py:201-425 (vertices / save_img_path dropped: unused by the arithmetic).
    Returns dict(atlas_img[A,A,3], shrinked[V,A,A], point_view_ids[P], points_atlas_pixel_coord[P,2],
    points[P,3], atlas_painted_mask[A,A], visibility[A,A,V], per_view_pixel[V,P,2])."""
    inpainted_images = np.asarray(inpainted_images, F32)
    res = mask.shape[1]
    V = len(cams)
    per_pixel_mask = np.asarray(mask)[0, :, :, 0].astype(bool)
    coords = np.argwhere(per_pixel_mask)                       # row-major (row, col) == reference order
    points = np.asarray(gb_pos, F32)[0][per_pixel_mask]
    P = points.shape[0]
    tp = np.zeros((V, P, 3), F32)
    for i, cam in enumerate(cams):
        tp[i] = cam.transform(points)
    depths = tp[..., 2]
    uv = tp[..., :2]
    pad9 = F32(1 - 2 * padding)
    uv = (uv - np.asarray(uv_centers, F32)) / np.asarray(uv_scales, F32)
    uv_ns = uv.copy()
    uv = uv * np.asarray(inpaint_scale_factors, F32)[:, None, None]
    uv = uv * pad9 + F32(0.5)
    uv_ns = uv_ns * pad9 + F32(0.5)
    vis, _ = point_validation_by_depth(cam_res, uv_ns, depths, mesh_normalized_depths, offset=0.0001)
    vis_AAV = np.zeros((res, res, V), bool)
    vis_AAV[per_pixel_mask] = vis.T
    kernel_sizes = list(edge_dilate_kernels) * (res // 256)    # list repetition, unproject.py:289
    per_kernel = shrink_visibility(per_pixel_mask, vis_AAV, kernel_sizes)       # [K,V,A,A]
    fid = np.asarray(per_atlas_pixel_face_id)[0]
    normals = np.asarray(f_normals, F32)[fid][per_pixel_mask]
    bd = np.asarray(base_dirs, F32)
    sim = np.stack([(normals[:, 0] * bd[v, 0] + normals[:, 1] * bd[v, 1]) + normals[:, 2] * bd[v, 2]
                    for v in range(V)], 1).astype(F32)
    pix = uv * F32(view_img_res)
    pix = np.clip(pix, F32(0), F32(view_img_res - 1))
    with np.errstate(invalid='ignore'):
        pix = pix.astype(np.int64)
    pix = np.stack([pix[:, :, 1], pix[:, :, 0]], -1)            # (row, col)
    shr = per_kernel[0]
    cand = shr[:, per_pixel_mask].T.copy()                      # [P,V]
    for i in range(1, len(edge_dilate_kernels)):
        left = cand.sum(1) < 1
        shr = per_kernel[i]
        cand[left] |= shr[:, per_pixel_mask].T[left]
    if complete_unseen_by_projection:
        left = cand.sum(1) < 1
        cand[left] |= vis.T[left]
    w = softmax_rows_f32(sim)
    w[~cand] = F32(-100)
    view_ids = w.argmax(1).astype(np.int64)
    if not complete_unseen_by_projection:
        view_ids[cand.sum(1) < 1] = -100
    atlas = np.zeros((res, res, 3), F32)
    painted = np.zeros((res, res), bool)
    for i in range(V):
        sel = view_ids == i
        img = inpainted_images[i][:, ::-1, :].transpose(1, 2, 0)
        atlas[coords[sel, 0], coords[sel, 1]] = img[pix[i][sel, 0], pix[i][sel, 1]]
        painted[coords[sel, 0], coords[sel, 1]] = True
    return dict(atlas_img=atlas, shrinked=shr, point_view_ids=view_ids, points_atlas_pixel_coord=coords,
                points=points, atlas_painted_mask=painted, visibility=vis_AAV, per_view_pixel=pix, sim=sim,
                per_kernel=per_kernel)
