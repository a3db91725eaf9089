"""Host-side mirror of the reference's pointdreamer/ours_utils.py for the texturing hot path.

Same function names, argument meaning, return order and tensor contracts as the reference
(/root/reference/pointdreamer/ours_utils.py) so that demo.colorize_one_mesh-style callers switch by
changing the import; every function routes to one C-ABI entry point of libpdhip.so (include/pdhip.h).
Torch is used for device memory and streams only.  There is no CPU path: CPU tensors raise.
"""
import os
import ctypes as C
import numpy as np
import torch

from . import _lib
from ._lib import ptr, as_u8, stream, check
from .camera_utils import stack_params
from . import io_utils


def _dev(t):
    if not t.is_cuda:
        raise _lib.PdhipError("expected a GPU tensor; pointdreamer_amd has no CPU path")
    return t.device


def crop_params(V, dev, uv_centers, uv_scales, padding, inpaint_scale_factors):
    """The crop / rescale parameters in the one form the kernels take: centres [V,2], scales [V], padding (float), factors [V].
    With crop_img=False the reference hands python scalars around (ours_utils.py:132-136: centre 0, scale 2, padding 0) and
    `None` for the inpaint scale factors where no rescale happened (unproject.py:260-262); both broadcast like tensors there."""
    def vec(x, n, default):
        if x is None:
            x = default
        if torch.is_tensor(x):
            x = x.to(dev).float().reshape(-1)
            return (x.expand(n) if x.numel() == 1 else x.reshape(n)).contiguous()
        return _lib.const_vec(n, x, dev)
    if torch.is_tensor(uv_centers) and uv_centers.numel() == 2 * V:
        uvc = uv_centers.to(dev).float().reshape(V, 2).contiguous()
    else:
        uvc = vec(uv_centers, 2 * V, 0.0).reshape(V, 2)
    return uvc, vec(uv_scales, V, 2.0), float(0.0 if padding is None else padding), vec(inpaint_scale_factors, V, 1.0)


def transform_points(cams, pts):
    """cam.transform for a list of cameras: pts[M,3] -> [V,M,3] NDC (kaolin Camera.transform contract)."""
    L = _lib.lib()
    pts = pts.reshape(-1, 3).float().contiguous()
    V, M = len(cams), pts.shape[0]
    dev = _dev(pts)
    cp = stack_params(cams)
    pos = torch.empty((V, M, 4), device=dev)
    vuv = torch.empty((V, M, 2), device=dev)
    ws = torch.empty((4 * V,), dtype=torch.int32, device=dev)
    check(L.pdhip_project_points(ptr(cp), V, ptr(pts), M, None, 0, 0, 0.0, ptr(pos), ptr(vuv), None, None, None, None,
                                 ptr(ws), stream()), 'pdhip_project_points')
    return pos[..., :3]


def get_rendered_hard_mask_and_face_idx_batch(cams, vertices, faces, points, glctx=None, rescale=True, padding=0.05):
    """ours_utils.py:93-150.  Returns, in the reference's order:
    hard_masks[V,R,R] bool, face_idxs[V,R,R] int64, mesh_normalized_depths[V,R,R] f32, vertice_uvs[V,Vn,2],
    uv_centers[V,1,2], uv_scales[V,1,1], padding, point_uvs[V,N,2], point_depths[V,N].  `glctx` is ignored."""
    L = _lib.lib()
    dev = _dev(vertices)
    vertices = vertices.float().contiguous()
    points = points.float().contiguous()
    faces32 = _lib.memo(faces, 'int32', lambda f: f.to(torch.int32).contiguous())
    V, Vn, N, F = len(cams), vertices.shape[0], points.shape[0], faces32.shape[0]
    R = int(cams[0].height)
    cp = stack_params(cams)
    pos = torch.empty((V, Vn, 4), device=dev)
    vuv = torch.empty((V, Vn, 2), device=dev)
    uvc = torch.empty((V, 1, 2), device=dev)
    uvs = torch.empty((V, 1, 1), device=dev)
    puv = torch.empty((V, N, 2), device=dev)
    pdep = torch.empty((V, N), device=dev)
    ws = torch.empty((4 * V,), dtype=torch.int32, device=dev)
    check(L.pdhip_project_points(ptr(cp), V, ptr(vertices), Vn, ptr(points), N, 1 if rescale else 0, float(padding),
                                 ptr(pos), ptr(vuv), ptr(uvc), ptr(uvs), ptr(puv), ptr(pdep), ptr(ws), stream()),
          'pdhip_project_points')
    ws_bytes = L.pdhip_raster_mesh_ws_bytes(V, F, R)          # face setups of the LDS-tiled path (meshes up to 65 536 faces) or z keys
    zkey = torch.empty(((ws_bytes + 7) // 8,), dtype=torch.int64, device=dev)
    hard = torch.empty((V, R, R), dtype=torch.bool, device=dev)
    fidx = torch.empty((V, R, R), dtype=torch.int64, device=dev)
    depth = torch.empty((V, R, R), device=dev)
    check(L.pdhip_raster_mesh_ws(ptr(pos), V, Vn, ptr(faces32), F, R, ptr(zkey), zkey.numel() * 8, ptr(as_u8(hard)), ptr(fidx), ptr(depth),
                                 stream()), 'pdhip_raster_mesh')
    if rescale:
        return hard, fidx, depth, vuv, uvc, uvs, padding, puv, pdep
    return hard, fidx, depth, vuv, 0, 2, 0, puv, pdep


def resize_masks(hard_masks, res):
    """demo.py:103-104: transforms.Resize((res,res))(mask.float()).bool() on [V,R,R] masks."""
    L = _lib.lib()
    hard_masks = hard_masks.contiguous()
    V, H, W = hard_masks.shape
    out = torch.empty((V, res, res), dtype=torch.bool, device=_dev(hard_masks))
    check(L.pdhip_resize_mask(ptr(as_u8(hard_masks)), V, H, W, ptr(as_u8(out)), res, res, stream()), 'pdhip_resize_mask')
    return out


def get_point_validation_by_depth(cam_res, point_uvs, point_depths, mesh_depths, offset=0, vis=False):
    """ours_utils.py:153-202 -> (visibility[V,N] bool, point_pixels[V,N,2] int64 (row,col))."""
    L = _lib.lib()
    dev = _dev(point_uvs)
    point_uvs = point_uvs.float().contiguous()
    point_depths = point_depths.float().contiguous()
    mesh_depths = mesh_depths.float().contiguous()
    V, N = point_depths.shape
    visb = torch.empty((V, N), dtype=torch.bool, device=dev)
    pix = torch.empty((V, N, 2), dtype=torch.int64, device=dev)
    check(L.pdhip_point_visibility(int(cam_res), ptr(point_uvs), ptr(point_depths), ptr(mesh_depths), V, N, float(offset),
                                   ptr(as_u8(visb)), ptr(pix), stream()), 'pdhip_point_visibility')
    return visb, pix


def get_point_validation_and_pixels(cam_res, point_uvs, point_depths, mesh_depths, res, offset=0):
    """get_point_validation_by_depth (its visibility) and get_point_pixels(point_uvs, res) in one pass over the points."""
    L = _lib.lib()
    dev = _dev(point_uvs)
    point_uvs = point_uvs.float().contiguous()
    point_depths = point_depths.float().contiguous()
    mesh_depths = mesh_depths.float().contiguous()
    V, N = point_depths.shape
    visb = torch.empty((V, N), dtype=torch.bool, device=dev)
    pix = torch.empty((V, N, 2), dtype=torch.int64, device=dev)
    check(L.pdhip_point_visibility_pixels(int(cam_res), ptr(point_uvs), ptr(point_depths), ptr(mesh_depths), V, N, float(offset),
                                          ptr(as_u8(visb)), int(res), ptr(pix), stream()), 'pdhip_point_visibility_pixels')
    return visb, pix


def get_point_pixels(point_uvs, res):
    """demo.py:121-125."""
    L = _lib.lib()
    point_uvs = point_uvs.float().contiguous()
    V, N = point_uvs.shape[:2]
    pix = torch.empty((V, N, 2), dtype=torch.int64, device=_dev(point_uvs))
    check(L.pdhip_point_pixels(ptr(point_uvs), V, N, int(res), ptr(pix), stream()), 'pdhip_point_pixels')
    return pix


def get_point_validation_by_o3d(points, eye_positions=None, hidden_point_removal_radius=None):
    """ours_utils.py:204-225 (Open3D hidden_point_removal per view) -> [V,N] bool."""
    from .hpr import hidden_point_removal
    return hidden_point_removal(points, eye_positions, hidden_point_removal_radius)


def refine_point_validation(cam_RTs, cam_K, res, hard_masks, point_validation, point_uvs, points, save_path, view_offset=0):
    """ours_utils.py:227-305 (`refine_point_validation_by_remove_abnormal_depth`, off in the shipped configs): per view, the camera-space
    depth z of the visible points is painted into a res x res map, filled to a dense map from the nearest painted pixel, and small
    foreground regions that are BRIGHTER (farther) than their surroundings -- points of the far side showing through -- are found
    by `utils_2d.detect_abnormal_bright_spots_in_gray_img`; visible points that land on such a region lose their visibility.
    cam_RTs [V,3,4] world -> camera, hard_masks [V,h,w] bool, point_validation [V,N] bool, point_uvs [V,N,2], points [N,3].
    The pixel arithmetic, the mask resize and the nearest fill run in the HIP kernels of rows P3 / P2b / I0; the blob test is host
    code as in the reference.  z = (x r20 + y r21) + z r22 + t2 in float32, in that order (the reference's torch.matmul leaves the
    summation order to the BLAS); several points on one pixel: the last one in point order wins (torch's CPU index_put order;
    on CUDA the reference's winner is unspecified).  save_path: directory for the `{i}_depth.png` panels, or None; view_offset: index of
    the first view in those names (view-parallel shards)."""
    from . import utils_2d
    dev = _dev(point_uvs)
    V, N = point_validation.shape
    pix = get_point_pixels(point_uvs, res)                                      # [V,N,2] (row, col), clipped
    fg = resize_masks(hard_masks, res)                                          # transforms.Resize((res, res)) then .astype(bool)
    RT = np.asarray(cam_RTs.detach().cpu() if torch.is_tensor(cam_RTs) else cam_RTs).astype(np.float32)
    pts = points.detach().float().cpu().numpy()
    zs_h = ((pts[None, :, 0] * RT[:, 2, 0:1] + pts[None, :, 1] * RT[:, 2, 1:2]) + pts[None, :, 2] * RT[:, 2, 2:3]) + RT[:, 2, 3:4]   # [V,N] f32
    pix_h, val_h, fg_h = pix.cpu().numpy(), point_validation.cpu().numpy().astype(bool), fg.cpu().numpy()
    sparse = np.full((V, res, res), -100.0, dtype=np.float32)
    for i in range(V):
        rc, zv = pix_h[i][val_h[i]], zs_h[i][val_h[i]]
        # duplicates: the LAST point in point order wins, made explicit (numpy leaves the winner of a repeated fancy index unspecified):
        # first occurrence in the reversed order = last occurrence
        lin = rc[:, 0].astype(np.int64) * res + rc[:, 1]
        _, first_rev = np.unique(lin[::-1], return_index=True)
        keep = len(lin) - 1 - first_rev
        sparse[i].reshape(-1)[lin[keep]] = zv[keep]
    sp = torch.from_numpy(sparse).to(dev)
    dense = nearest_fill(sp[:, None], sp != -100.0, 'CHW')[:, 0].cpu().numpy()  # naive_inpainting(method='nearest') per view
    new_val = val_h.copy()
    for i in range(V):
        path = None
        if save_path is not None:
            os.makedirs(save_path, exist_ok=True)
            path = os.path.join(save_path, f'{i + view_offset}_depth.png')
        abnormal = utils_2d.detect_abnormal_bright_spots_in_gray_img(dense[i], fg_h[i], save_path=path, min_for_norm=0.5, max_for_norm=2.5,
                                                                    edge_thresh=25, pixel_num_thresh=2000, area_expand_thresh=5,
                                                                    area_same_color_thres=5, brighter_thresh=5)
        rc = pix_h[i][val_h[i]]
        new_val[i][val_h[i]] = ~abnormal[rc[:, 0], rc[:, 1]]
    return torch.from_numpy(new_val).to(dev)


def get_sparse_images(point_pixels, colors, point_validation, hard_masks, save_path, view_num, res, point_size,
                      edge_point_size, mask_ratio_thresh, view_offset=0):
    """ours_utils.py:848-882 -> sparse_imgs[V,3,r,r], hard_mask0s, hard_mask2s, scale_factors[V].
    view_offset: index of the first view in the file names (a rank that owns views [lo, hi) of a shape)."""
    L = _lib.lib()
    dev = _dev(point_pixels)
    point_pixels = point_pixels.to(torch.int64).contiguous()
    colors = colors.float().contiguous()
    point_validation = point_validation.contiguous()
    hard_masks = hard_masks.contiguous()
    V, N = point_pixels.shape[:2]
    assert V == view_num and hard_masks.shape[-1] == res
    sparse = torch.empty((V, 3, res, res), device=dev)
    m0 = torch.empty_like(sparse)
    m2 = torch.empty_like(sparse)
    sf = torch.empty((V,), device=dev)
    ws = torch.empty((L.pdhip_sparse_views_ws_bytes(V, N, res),), dtype=torch.uint8, device=dev)
    check(L.pdhip_sparse_views(ptr(point_pixels), ptr(colors), ptr(as_u8(point_validation)), ptr(as_u8(hard_masks)),
                               V, N, res, int(point_size), int(edge_point_size), float(mask_ratio_thresh),
                               ptr(sparse), ptr(m0), ptr(m2), ptr(sf), None, ptr(ws), stream()), 'pdhip_sparse_views')
    if save_path is not None:
        os.makedirs(save_path, exist_ok=True)
        for i in range(V):
            save_mask = (m0[i][0] * m2[i][0]).unsqueeze(0)
            k = i + view_offset
            io_utils.save_CHW_RGBA_img(torch.cat([sparse[i], save_mask]), os.path.join(save_path, f'{k}_sparse.png'))
            io_utils.save_CHW_RGB_img(m0[i], os.path.join(save_path, f'{k}_mask0.png'))
            io_utils.save_CHW_RGB_img(m2[i], os.path.join(save_path, f'{k}_mask2.png'))
    return sparse, m0, m2, sf


def nearest_fill(img, site_mask, layout='CHW'):
    """Batched exact nearest-site fill.  img [B,C,H,W] (layout 'CHW') or [B,H,W,C] ('HWC') float32;
    site_mask [B,H,W] bool/uint8 or float (site iff != 0), or [B,K,H,W] of which plane 0 of every image is used in place
    (the kernel takes a batch stride: no copy of `mask2[:, 0]`)."""
    L = _lib.lib()
    img = img.float().contiguous()
    dev = _dev(img)
    if layout == 'CHW':
        B, Cn, H, W = img.shape
        bs, cs, ps = Cn * H * W, H * W, 1
    else:
        B, H, W, Cn = img.shape
        bs, cs, ps = H * W * Cn, 1, Cn
    site_mask = site_mask.contiguous()
    # the mask is read plane 0 of image b at b * mstride: [B,H,W] or [B,K,H,W] only (a [B,H,W,1] mask would be read with the wrong stride)
    if site_mask.dim() not in (3, 4) or site_mask.shape[0] != B or tuple(site_mask.shape[-2:]) != (H, W):
        raise _lib.PdhipError(f"site mask must be [B,H,W] or [B,K,H,W] with B={B}, H={H}, W={W} (got {tuple(site_mask.shape)})")
    mstride = H * W * (site_mask.shape[1] if site_mask.dim() == 4 else 1)
    is_f32 = 1 if site_mask.dtype == torch.float32 else 0
    if not is_f32:
        site_mask = as_u8(site_mask)
        if site_mask.dtype != torch.uint8:
            raise _lib.PdhipError("site mask must be bool, uint8 or float32")
    out = torch.empty_like(img)
    ws = torch.empty((L.pdhip_nearest_fill_ws_ints(B, H, W),), dtype=torch.int32, device=dev)
    check(L.pdhip_nearest_fill(ptr(img), ptr(out), B, Cn, H, W, bs, cs, ps, ptr(site_mask), is_f32, mstride, ptr(ws),
                               stream()), 'pdhip_nearest_fill')
    return out


def linear_fill(img, site_mask, return_triangles=False):
    """Batched Delaunay-linear fill (scipy griddata method='linear' per image): img [B,C,H,W] float32, site_mask [B,H,W]
    bool / uint8 / float (site iff != 0).  Unknown pixels outside the convex hull of the sites become NaN, as with scipy.
    return_triangles: also the [B,H,W,3] int32 site indices of the triangle used per pixel (tests)."""
    L = _lib.lib()
    img = img.float().contiguous()
    dev = _dev(img)
    B, Cn, H, W = img.shape
    site_mask = site_mask.contiguous()
    is_f32 = 1 if site_mask.dtype == torch.float32 else 0
    if not is_f32:
        site_mask = as_u8(site_mask)
        if site_mask.dtype != torch.uint8:
            raise _lib.PdhipError("site mask must be bool, uint8 or float32")
    out = torch.empty_like(img)
    ws = torch.empty((L.pdhip_linear_fill_ws_bytes(B, H, W),), dtype=torch.uint8, device=dev)
    tri = torch.empty((B, H, W, 3), dtype=torch.int32, device=dev) if return_triangles else None
    check(L.pdhip_linear_fill(ptr(img), ptr(out), B, Cn, H, W, ptr(site_mask), is_f32, H * W, ptr(ws), ptr(tri, allow_none=True),
                              stream()), 'pdhip_linear_fill')
    # the round-cap counter is read on every path ('linear' is a synchronous, tens-of-milliseconds method: the 4-byte read is free);
    # such pixels carry the interpolant of a containing, not necessarily Delaunay, triangle -- never a silent NaN
    n = C.c_int(0)
    check(L.pdhip_linear_fill_unresolved(ptr(ws), B, H, W, C.byref(n), stream()), 'pdhip_linear_fill_unresolved')
    if n.value:
        if return_triangles:
            raise _lib.PdhipError(f"linear_fill: {n.value} pixels hit the round cap")
        import warnings
        warnings.warn(f"linear_fill: {n.value} pixels hit the round cap of the Delaunay search (interpolated in a containing triangle)")
    return (out, tri) if return_triangles else out


def naive_inpainting(img, no_need_inpaint_mask2, method='linear'):
    """ours_utils.py:610-643.  img[C,H,W], mask2[C,H,W] (channel 0 used) -> [C,H,W] float32 on the GPU.
    'nearest': exact nearest site (tie rule in DESIGN.md); 'linear': Delaunay-linear (csrc/linear.hip)."""
    if method not in ('nearest', 'linear'):
        raise ValueError(f"naive_inpainting: method {method!r} (scipy griddata knows 'cubic' too; the reference never uses it)")
    m = no_need_inpaint_mask2[0:1].contiguous()
    if m.dtype not in (torch.float32, torch.bool, torch.uint8):
        m = m.float()
    if method == 'linear':
        return linear_fill(img.unsqueeze(0), m)[0]
    return nearest_fill(img.unsqueeze(0), m, 'CHW')[0]


def save_inpainted_images(out, hard_mask0s, save_path, view_num, method, view_offset=0):
    """`{k}_inpainted.png`: RGBA with alpha = mask0 for DDNM (ours_utils.py:924-928), RGB for nearest (:939-941)."""
    os.makedirs(save_path, exist_ok=True)
    for i in range(view_num):
        if method == 'DDNM_inpaint':
            rgba = torch.cat([out[i], hard_mask0s[i][0].unsqueeze(0)])
            io_utils.save_CHW_RGBA_img(rgba, os.path.join(save_path, f'{i + view_offset}_inpainted.png'))
        else:
            io_utils.save_CHW_RGB_img(out[i], os.path.join(save_path, f'{i + view_offset}_inpainted.png'))


def load_inpainted_images(save_path, view_num, device):
    """demo.py:138-147: when every `{i}_inpainted.png` of a shape is already on disk the reference loads them instead of
    inpainting again (its resume surface: the 8-bit RGB of the saved files).  Returns [V,3,r,r] float32 or None."""
    if save_path is None:
        return None
    paths = [os.path.join(save_path, f'{i}_inpainted.png') for i in range(view_num)]
    if not all(os.path.exists(p) for p in paths):
        return None
    io_utils.flush()                                  # a queued write of the same files must have landed
    return torch.stack([io_utils.load_CHW_RGB_img(p) for p in paths], 0).to(device)


def get_inpainted_images(sparse_imgs, hard_mask0s, hard_mask2s, save_path, inpainter, view_num, method='linear',
                         first_key=None, advance=None, view_offset=0):
    """ours_utils.py:884-951 -> inpainted[V,3,r,r].  first_key / advance: the noise keys of these views (Inpainter.inpaint_views)."""
    if method == 'DDNM_inpaint':
        out = inpainter.inpaint_views(sparse_imgs, hard_mask2s[:, 0].contiguous(), first_key=first_key, advance=advance)
    elif method == 'nearest':
        out = nearest_fill(sparse_imgs, hard_mask2s, 'CHW')                  # (plane 0 of every view's mask, read in place)
    elif method == 'linear':
        out = linear_fill(sparse_imgs, hard_mask2s[:, 0].contiguous())
    else:
        raise NotImplementedError(f"texture_gen_method={method!r} is not built (DDNM_inpaint | nearest | linear)")
    if save_path is not None:
        save_inpainted_images(out, hard_mask0s, save_path, view_num, method, view_offset)
    return out
