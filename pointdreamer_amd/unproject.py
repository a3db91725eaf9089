"""Host-side mirror of the reference's pointdreamer/unproject.py (rows Uq1-Uq5, N1-N3): same names,
argument order and return tuples (/root/reference/pointdreamer/unproject.py:201-425, 429-475, 480-504);
each routes to libpdhip.so.  The debug PNG triptychs the reference always writes
(unproject.py:459-474, `others/shrink_per_view_edge/{v}.png`) are produced by save_shrink_triptychs when a save path is given."""
import ctypes as C
import torch

from . import _lib
from ._lib import ptr, as_u8, stream, check
from .camera_utils import stack_params
from .ours_utils import nearest_fill, _dev, crop_params


def texel_visibility(cams, gb_pos, mask, uv_centers, uv_scales, padding, mesh_normalized_depths, cam_res, offset=0.0001):
    """Uq1+Uq2 (unproject.py:219-284) -> visibility[V,A,A] bool."""
    L = _lib.lib()
    dev = _dev(gb_pos)
    V = len(cams)
    A = mask.shape[1]
    cp = stack_params(cams)
    gb = gb_pos[0].float().contiguous()
    m = as_u8(mask[0, :, :, 0].contiguous())
    uvc, uvs, padding, _ = crop_params(V, dev, uv_centers, uv_scales, padding, None)
    md = mesh_normalized_depths.float().contiguous()
    vis = torch.empty((V, A, A), dtype=torch.bool, device=dev)
    check(L.pdhip_texel_visibility(ptr(cp), V, ptr(gb), ptr(m), A, ptr(uvc), ptr(uvs), float(padding), ptr(md),
                                   int(cam_res), float(offset), ptr(as_u8(vis)), stream()), 'pdhip_texel_visibility')
    return vis


def get_shrinked_per_view_per_pixel_visibility_torch(per_pixel_mask, per_atlas_pixel_per_view_visibility,
                                                     kernel_sizes=[21], save_path=None):
    """unproject.py:429-475.  per_pixel_mask[A,A] bool, visibility [A,A,V] bool (reference layout)
    -> [K,V,A,A] bool."""
    vis = per_atlas_pixel_per_view_visibility.permute(2, 0, 1).contiguous()
    return shrink_visibility(per_pixel_mask, vis, kernel_sizes)


def shrink_visibility(per_pixel_mask, vis_VAA, kernel_sizes):
    L = _lib.lib()
    dev = _dev(vis_VAA)
    V, A, _ = vis_VAA.shape
    ks = [int(k) for k in kernel_sizes]
    K = 1 if ks[0] == 0 else len(ks)
    arr = (C.c_int32 * len(ks))(*ks)
    out = torch.empty((K, V, A, A), dtype=torch.bool, device=dev)
    ws = torch.empty((2, V, A, A), dtype=torch.uint8, device=dev)
    check(L.pdhip_nbf_shrink(ptr(as_u8(per_pixel_mask.contiguous())), ptr(as_u8(vis_VAA.contiguous())), V, A, arr, K,
                             ptr(as_u8(out)), ptr(ws), stream()), 'pdhip_nbf_shrink')
    return out


def save_shrink_triptychs(per_pixel_mask, vis_VAA, kernel, save_path, view_offset=0):
    """`{save_path}/{v}.png` exactly as unproject.py:459-474 composes them (device composition, native PNG encode on the
    io_utils write queue)."""
    import os
    from . import io_utils
    L = _lib.lib()
    dev = _dev(vis_VAA)
    V, A, _ = vis_VAA.shape
    out = torch.empty((V, A, 3 * A + 20, 3), dtype=torch.uint8, device=dev)
    ws = torch.empty(((2 * V + 1) * A * A,), dtype=torch.uint8, device=dev)
    check(L.pdhip_nbf_triptych(ptr(as_u8(per_pixel_mask.contiguous())), ptr(as_u8(vis_VAA.contiguous())), V, A, int(kernel),
                               ptr(out), ptr(ws), stream()), 'pdhip_nbf_triptych')
    os.makedirs(save_path, exist_ok=True)
    host, wait = io_utils._host_u8(out)
    for v in range(V):
        io_utils.save_HWC_u8_img(host[v], os.path.join(save_path, f'{v + view_offset}.png'), wait=wait)


def per_view_visibility(cams, cam_res, gb_pos, mask, uv_centers, uv_scales, padding, mesh_normalized_depths, edge_dilate_kernels,
                        save_path=None, view_offset=0):
    """Uq1-Uq2 + N1-N3 of the given views (each view is independent of the others -- the part of unproject() a rank owns for its
    views under view-parallel sharding): visibility [V,A,A] bool, shrunk visibility per kernel level [K,V,A,A] bool."""
    dev = _dev(gb_pos)
    V = len(cams)
    A = mask.shape[1]
    uvc, uvs, padding, _ = crop_params(V, dev, uv_centers, uv_scales, padding, None)
    vis = texel_visibility(cams, gb_pos, mask, uvc, uvs, padding, mesh_normalized_depths, cam_res)
    per_pixel_mask = mask[0, :, :, 0].contiguous()
    kernel_sizes = list(edge_dilate_kernels) * (A // 256)          # list repetition, unproject.py:289
    if len(kernel_sizes) == 0:
        raise _lib.PdhipError("atlas resolution < 256 gives an empty kernel list in the reference (IndexError); use A >= 256")
    # the reference shrinks with every entry of the repeated list but only levels 0..len(edge_dilate_kernels)-1 are ever
    # consulted (unproject.py:324-346): compute exactly those
    K = 1 if int(kernel_sizes[0]) == 0 else min(len(edge_dilate_kernels), len(kernel_sizes))
    per_kernel = shrink_visibility(per_pixel_mask, vis, kernel_sizes[:K])
    if save_path is not None and int(kernel_sizes[0]) != 0:
        save_shrink_triptychs(per_pixel_mask, vis, int(kernel_sizes[-1]), save_path, view_offset)
    return vis, per_kernel


def blend_views(inpainted_images, f_normals, view_img_res, cams, base_dirs, gb_pos, mask, per_atlas_pixel_face_id, uv_centers,
                uv_scales, padding, inpaint_scale_factors, vis, per_kernel, complete_unseen_by_projection=False):
    """Uq3-Uq4 over ALL views (unproject.py:298-400): per-texel view selection + colour fetch.
    Returns atlas[A,A,3], view_ids[A,A] int32, painted[A,A] bool."""
    L = _lib.lib()
    dev = _dev(gb_pos)
    V = len(cams)
    A = mask.shape[1]
    K = per_kernel.shape[0]
    uvc, uvs, padding, sf = crop_params(V, dev, uv_centers, uv_scales, padding, inpaint_scale_factors)
    per_pixel_mask = mask[0, :, :, 0].contiguous()
    cp = stack_params(cams)
    gb = gb_pos[0].float().contiguous()
    fid = per_atlas_pixel_face_id[0].to(torch.int64).contiguous()
    fn = f_normals.float().contiguous()
    bd = base_dirs.float().contiguous()
    img = inpainted_images.float().contiguous()
    atlas = torch.empty((A, A, 3), device=dev)
    painted = torch.empty((A, A), dtype=torch.bool, device=dev)
    view_ids = torch.empty((A, A), dtype=torch.int32, device=dev)
    check(L.pdhip_view_select_blend(ptr(cp), V, ptr(gb), ptr(as_u8(per_pixel_mask)), ptr(fid), A, ptr(fn), ptr(bd),
                                    ptr(uvc), ptr(uvs), float(padding), ptr(sf), ptr(as_u8(per_kernel.contiguous())), K,
                                    ptr(as_u8(vis.contiguous())), 1 if complete_unseen_by_projection else 0, ptr(img),
                                    int(view_img_res), ptr(atlas), ptr(as_u8(painted)), ptr(view_ids), stream()),
          'pdhip_view_select_blend')
    return atlas, view_ids, painted


def unproject_dense(inpainted_images, f_normals, view_img_res, cams, cam_res, base_dirs, gb_pos, mask,
                    per_atlas_pixel_face_id, uv_centers, uv_scales, padding, inpaint_scale_factors,
                    mesh_normalized_depths, edge_dilate_kernels, complete_unseen_by_projection=False, save_img_path=None,
                    vis_and_shrunk=None):
    """The whole of unproject() in its dense [A,A] form (no compaction, no host sync).
    Returns atlas[A,A,3], shrinked[V,A,A] bool (last level), view_ids[A,A] int32, painted[A,A] bool, vis[V,A,A].
    vis_and_shrunk: (vis, per_kernel) already computed (gathered from the ranks that own the views)."""
    if uv_scales is None or uv_centers is None or inpaint_scale_factors is None or padding is None:
        # unproject.py:260-262: uv = xy*0.5+0.5  ==  centre 0, scale 2, padding 0, factor 1
        uv_centers, uv_scales, inpaint_scale_factors, padding = None, None, None, None
    if vis_and_shrunk is None:
        import os
        vis, per_kernel = per_view_visibility(
            cams, cam_res, gb_pos, mask, uv_centers, uv_scales, padding, mesh_normalized_depths, edge_dilate_kernels,
            None if save_img_path is None else os.path.join(save_img_path, 'shrink_per_view_edge'))
    else:
        vis, per_kernel = vis_and_shrunk
    atlas, view_ids, painted = blend_views(inpainted_images, f_normals, view_img_res, cams, base_dirs, gb_pos, mask,
                                           per_atlas_pixel_face_id, uv_centers, uv_scales, padding, inpaint_scale_factors, vis,
                                           per_kernel, complete_unseen_by_projection)
    return atlas, per_kernel[per_kernel.shape[0] - 1], view_ids, painted, vis


def compact_texels(gb_pos, mask, view_ids):
    """Row-major compaction of the chart texels (unproject.py:223-233): points[P,3], coords[P,2], view ids[P]."""
    L = _lib.lib()
    dev = _dev(gb_pos)
    A = mask.shape[1]
    gb = gb_pos[0].float().contiguous()
    m = as_u8(mask[0, :, :, 0].contiguous())
    n = A * A
    points = torch.empty((n, 3), device=dev)
    coords = torch.empty((n, 2), dtype=torch.int64, device=dev)
    pvid = torch.empty((n,), dtype=torch.int64, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((A + 1,), dtype=torch.int32, device=dev)
    check(L.pdhip_compact_texels(ptr(gb), ptr(m), A, ptr(view_ids.contiguous()), ptr(points), ptr(coords), ptr(pvid),
                                 ptr(cnt), ptr(ws), stream()), 'pdhip_compact_texels')
    P = int(cnt.item())                                              # the one host sync (the reference syncs here too)
    return points[:P], coords[:P], pvid[:P]


def unproject(inpainted_images, vertices, f_normals, view_img_res, cams, cam_res, base_dirs, gb_pos, mask,
              per_atlas_pixel_face_id, uv_centers, uv_scales, padding, inpaint_scale_factors, mesh_normalized_depths,
              edge_dilate_kernels, save_img_path, complete_unseen_by_projection=False):
    """unproject.py:201-425, same 18 positional arguments and the same 6-tuple:
    atlas_img[A,A,3], shrinked_per_view_per_pixel_visibility[V,A,A], point_view_ids[P], points_atlas_pixel_coord[P,2],
    points[P,3], atlas_painted_mask[A,A]."""
    atlas, shr, view_ids, painted, _ = unproject_dense(
        inpainted_images, f_normals, view_img_res, cams, cam_res, base_dirs, gb_pos, mask, per_atlas_pixel_face_id,
        uv_centers, uv_scales, padding, inpaint_scale_factors, mesh_normalized_depths, edge_dilate_kernels,
        complete_unseen_by_projection, save_img_path)
    points, coords, pvid = compact_texels(gb_pos, mask, view_ids)
    return atlas, shr, pvid, coords, points, painted


def dilate_atlas(atlas_img, mask):
    """unproject.py:480-504: atlas_img[A,A,3], mask[1,A,A,1] bool -> [A,A,3] (float32 on the GPU)."""
    m = mask[..., 0].contiguous()
    return nearest_fill(atlas_img.unsqueeze(0), m, 'HWC')[0]


def unpainted_face_ids(per_atlas_pixel_face_id, atlas_painted_mask, num_faces):
    """demo.py:180-181: ids of the faces that own at least one chart texel no view painted (sorted, int64, on the host)."""
    import numpy as np
    L = _lib.lib()
    fid = per_atlas_pixel_face_id[0].to(torch.int64).contiguous()
    A = fid.shape[0]
    flags = torch.empty((int(num_faces),), dtype=torch.uint8, device=_dev(fid))
    check(L.pdhip_mark_unpainted_faces(ptr(fid), ptr(as_u8(atlas_painted_mask.contiguous())), A, int(num_faces), ptr(flags), stream()),
          'pdhip_mark_unpainted_faces')
    return np.nonzero(flags.cpu().numpy())[0].astype(np.int64)


def paint_invisible_areas_by_neighbors(vertices, faces, uvs, face_uv_idx, to_inpaint_face_id, atlas_img, atlas_inpainted_mask,
                                       use_atlas=True):
    """unproject.py:93-196, same arguments.  atlas_img [A,A,3] f32, atlas_inpainted_mask [A,A] bool (both on the GPU).
    Two rounds of midpoint subdivision of the unpainted faces (host numpy, as in the reference), per-vertex colour fetch,
    Jacobi neighbour averaging with the reference's loop control, scatter back, exact nearest fill.
    use_atlas=True -> atlas [A,A,3]; False -> (subdivided_vertices, subdivided_faces, vertex_colors)."""
    import numpy as np
    from . import mesh_utils as mu
    L = _lib.lib()
    dev = _dev(atlas_img)
    A = atlas_inpainted_mask.shape[1]
    to_np = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    sv, sf, su, sfu = to_np(vertices), to_np(faces), to_np(uvs), to_np(face_uv_idx)
    tif = to_np(to_inpaint_face_id).astype(np.int64)
    for _ in range(2):                      # unproject.py:111-114: the same index list both rounds (not re-mapped) -- kept
        sv, sf, su, sfu = mu.subdivide_with_uv(sv, sf, sfu, su, face_index=tif)
    V = len(sv)
    vert_uvs = torch.from_numpy(mu.vertex_uv_table(V, sf, sfu, su)).to(dev)
    rowptr, colidx = mu.neighbour_csr(V, sf)
    rowptr_d, colidx_d = torch.from_numpy(rowptr).to(dev), torch.from_numpy(colidx).to(dev)
    atlas = atlas_img.float().contiguous().clone()
    mask = as_u8(atlas_inpainted_mask.contiguous()).clone()
    texel = torch.empty((V, 2), dtype=torch.int32, device=dev)
    colors = torch.empty((V, 3), device=dev)
    count = torch.empty((V,), device=dev)
    check(L.pdhip_vertex_texel_fetch(ptr(vert_uvs), V, ptr(atlas), ptr(mask), A, ptr(texel), ptr(colors), ptr(count), stream()),
          'pdhip_vertex_texel_fetch')
    invalid_np = np.nonzero(count.cpu().numpy() == 0)[0].astype(np.int32)
    IV = len(invalid_np)
    invalid = torch.from_numpy(invalid_np).to(dev) if IV else torch.zeros((1,), dtype=torch.int32, device=dev)
    tmp = torch.empty((max(IV, 1) * 4,), device=dev)
    colored = torch.zeros((1,), dtype=torch.int32, device=dev)
    total, rounds, stage = V - IV, 0, "uncolored"
    while stage == "uncolored" or rounds > 0:            # unproject.py:159-178 (one host sync per round, as the reference)
        check(L.pdhip_neighbor_diffuse_round(ptr(rowptr_d), ptr(colidx_d), V, ptr(invalid), IV, ptr(colors), ptr(count), ptr(tmp),
                                             ptr(colored), stream()), 'pdhip_neighbor_diffuse_round')
        new_total = (V - IV) + int(colored.item())
        if new_total > total:
            total, rounds = new_total, rounds + 1
        else:
            stage, rounds = "colored", rounds - 1
        if rounds > 10000:
            break
    if not use_atlas:
        return torch.from_numpy(sv).to(dev), torch.from_numpy(sf).to(dev).long(), colors
    owner = torch.empty((A * A,), dtype=torch.int32, device=dev)
    check(L.pdhip_scatter_vertex_colors(ptr(texel), ptr(colors), V, ptr(atlas), ptr(mask), ptr(owner), A, stream()),
          'pdhip_scatter_vertex_colors')
    return nearest_fill(atlas.unsqueeze(0), mask.unsqueeze(0), 'HWC')[0]
