"""BASELINE configs[4] (a batch of shapes per GPU), round 4: S independent shapes (equal cloud and atlas sizes; meshes of ANY sizes since round 6,
padded by `stack`) through the texturing path with ONE
launch per stage for all of them (`pdhip_*_shapes`, include/pdhip.h) instead of S launches (or S HIP graphs) per stage.

Per-shape inputs are stacked: coords / colors [S,N,3], vertices [S,Vn,3], faces [S,F,3], f_normals [S,F,3], gb_pos [S,A,A,3],
mask [S,A,A,1], per_atlas_pixel_face_id [S,A,A]; the V cameras are shared by the shapes (demo.py builds them once per run).  Every
per-view array has S*V leading entries, view g = s * V + v.  The stages are the ones of pipeline._before_inpaint / _after_inpaint
(demo.py:93-129, 167-178) with the same arithmetic: the atlases equal colorize_one_mesh's shape by shape, bit for bit (tested).

Built here: texture_gen_method 'nearest' | 'linear' | 'DDNM_inpaint', complete_unseen_by='unproject', optimize_from=None, no per-view
files -- what `pipeline.colorize_meshes_batched` falls back from for anything else."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ptr, as_u8, stream, check
from . import ours_utils as ou
from .camera_utils import stack_params

_EYES = {}


def uniform(shapes):
    """Can these shape dicts (pipeline.colorize_meshes_batched) be stacked?  Equal cloud size N and atlas size A, every tensor of every shape
    on the SAME device (the caller checks that it is a CUDA device), and atlas maps with the leading singleton dimension stack() indexes away
    (gb_pos [1, A, A, 3], mask [1, A, A, 1], face ids [1, A, A]).  The MESHES may differ in vertex and face count (round 6, BASELINE
    configs[4]: real POCO meshes do): `stack` pads them -- see there.  Clouds of different sizes take the per-shape route (a padded point
    would take part in the largest-index-wins splat)."""
    def sig(sh):
        x = sh['xatlas']
        ts = (sh['coords'], sh['colors'], sh['vertices'], sh['faces'], sh['f_normals'], x['gb_pos'], x['mask'], x['per_atlas_pixel_face_id'])
        if not all(torch.is_tensor(t) for t in ts) or len({t.device for t in ts}) != 1:
            return None
        if not (x['gb_pos'].dim() == 4 and x['mask'].dim() == 4 and x['per_atlas_pixel_face_id'].dim() == 3 and
                x['gb_pos'].shape[0] == 1 and x['mask'].shape[0] == 1 and x['per_atlas_pixel_face_id'].shape[0] == 1):
            return None
        if not (sh['vertices'].dim() == 2 and sh['faces'].dim() == 2 and sh['f_normals'].shape == (sh['faces'].shape[0], 3) and
                sh['vertices'].shape[0] >= 1 and sh['faces'].shape[0] >= 1):
            return None
        return tuple((tuple(t.shape), t.device) for t in (sh['coords'], sh['colors'], x['gb_pos'], x['mask'], x['per_atlas_pixel_face_id']))
    if len(shapes) < 1:
        return False
    s0 = sig(shapes[0])
    return s0 is not None and all(sig(sh) == s0 for sh in shapes[1:])


def ragged(shapes):
    """True when the meshes of a stackable batch differ in vertex or face count (the stacked tensors then carry padding)."""
    return len({(sh['vertices'].shape[0], sh['faces'].shape[0]) for sh in shapes}) > 1


def stack(shapes):
    """Shape dicts -> the stacked tensors `colorize_shapes` takes (one copy of every input; callers that texture many batches of the
    same mesh / atlas keep the geometry part).  Meshes of different sizes are padded to the largest: vertices with copies of the shape's
    vertex 0 (the crop bounds of P1 are a min / max over the vertices: unchanged), faces with the degenerate triangle (0, 0, 0) (zero area:
    culled by the rasteriser's set-up, never a face id), face normals with zeros (indexed by real face ids only) -- every per-shape result
    stays bit-identical to colorize_one_mesh's (tested), at one launch per stage for the whole batch (the reference loops shape by shape:
    demo.py:455-462)."""
    vmax = max(sh['vertices'].shape[0] for sh in shapes)
    fmax = max(sh['faces'].shape[0] for sh in shapes)

    def pad_rows(t, n, fill_row):
        if t.shape[0] == n:
            return t
        return torch.cat([t, fill_row.to(t.dtype).reshape(1, -1).expand(n - t.shape[0], -1)], 0)
    cat = lambda f: torch.stack([f(sh) for sh in shapes], 0).contiguous()
    z3 = lambda t: torch.zeros((3,), dtype=t.dtype, device=t.device)
    return dict(coords=cat(lambda s: s['coords'].float()), colors=cat(lambda s: s['colors'].float()),
                vertices=cat(lambda s: pad_rows(s['vertices'].float(), vmax, s['vertices'][0].float())),
                faces=cat(lambda s: pad_rows(s['faces'].to(torch.int32), fmax, z3(s['faces'].to(torch.int32)))),
                f_normals=cat(lambda s: pad_rows(s['f_normals'].float(), fmax, z3(s['f_normals'].float()))),
                gb_pos=cat(lambda s: s['xatlas']['gb_pos'][0].float()),
                mask=cat(lambda s: s['xatlas']['mask'][0]), face_id=cat(lambda s: s['xatlas']['per_atlas_pixel_face_id'][0].to(torch.int64)))


def _eyes(eye_positions, S, dev):
    eyes_h = np.ascontiguousarray(np.tile(np.asarray(eye_positions, np.float64).reshape(-1, 3), (S, 1)))
    key = (eyes_h.tobytes(), str(dev))
    e = _EYES.get(key)
    if e is None:
        if len(_EYES) > 64:
            _EYES.clear()
        e = _EYES[key] = torch.from_numpy(eyes_h).to(dev).contiguous()
        _lib._settle(e)
    return e


def colorize_shapes(st, camera_info, view_num, res, cam_res, inpainter=None, texture_gen_method='nearest', point_size=1,
                    edge_point_size=1, crop_img=True, crop_padding=0.05, mask_ratio_thresh=0.82, edge_dilate_kernels=(21,),
                    point_validation_by_o3d=True, hidden_point_removal_radius=100, return_intermediates=False):
    """st: dict of stacked tensors (`stack`).  Returns the dilated atlases [S,A,A,3] (and the intermediates on request)."""
    L = _lib.lib()
    coords, colors, vertices, faces = st['coords'], st['colors'], st['vertices'], st['faces']
    dev = coords.device
    cams = camera_info['cams']
    V = len(cams)
    assert V == view_num
    S, N = coords.shape[0], coords.shape[1]
    Vn, F = vertices.shape[1], faces.shape[1]
    A = st['mask'].shape[1]
    R = int(cams[0].height)
    G = S * V
    if G > 64:
        raise _lib.PdhipError(f"colorize_shapes: at most 64 views per call (S * V = {G}); split the batch")
    cp = stack_params(cams)
    rescale = 1 if crop_img else 0
    with torch.no_grad():
        # ---- P1 + P2 (ours_utils.py:93-150)
        pos = torch.empty((G, Vn, 4), device=dev)
        vuv = torch.empty((G, Vn, 2), device=dev)
        uvc = torch.empty((G, 2), device=dev)
        uvs = torch.empty((G,), device=dev)
        puv = torch.empty((G, N, 2), device=dev)
        pdep = torch.empty((G, N), device=dev)
        mm = torch.empty((4 * G,), dtype=torch.int32, device=dev)
        check(L.pdhip_project_points_shapes(ptr(cp), V, S, ptr(vertices), Vn, ptr(coords), N, rescale, float(crop_padding), ptr(pos), ptr(vuv),
                                            ptr(uvc), ptr(uvs), ptr(puv), ptr(pdep), ptr(mm), stream()), 'pdhip_project_points_shapes')
        if not crop_img:                                  # ours_utils.py:132-136: centre 0, scale 2, padding 0
            uvc.zero_(); uvs.fill_(2.0)
        padding = float(crop_padding) if crop_img else 0.0
        ws_bytes = L.pdhip_raster_mesh_ws_bytes(G, F, R)
        zkey = torch.empty(((ws_bytes + 7) // 8,), dtype=torch.int64, device=dev)
        hard = torch.empty((G, R, R), dtype=torch.bool, device=dev)
        fidx = torch.empty((G, R, R), dtype=torch.int64, device=dev)
        depth = torch.empty((G, R, R), device=dev)
        check(L.pdhip_raster_mesh_shapes(ptr(pos), V, S, Vn, ptr(faces), F, R, ptr(zkey), zkey.numel() * 8, ptr(as_u8(hard)), ptr(fidx),
                                         ptr(depth), stream()), 'pdhip_raster_mesh_shapes')
        # ---- Uq1-Uq2 + N1-N3 (they need only the depth maps; queued on a side stream beside the hidden-point removal they gain nothing:
        # 2.309 against 2.303 ms per 8 shapes -- every kernel here already fills the chip)
        gb, mask, fid, fn = st['gb_pos'], st['mask'], st['face_id'], st['f_normals']
        m8 = as_u8(mask.reshape(S, A, A).contiguous())
        kernel_sizes = list(edge_dilate_kernels) * (A // 256)          # list repetition, unproject.py:289
        if len(kernel_sizes) == 0:
            raise _lib.PdhipError("atlas resolution < 256 gives an empty kernel list in the reference (IndexError); use A >= 256")
        K = 1 if int(kernel_sizes[0]) == 0 else min(len(edge_dilate_kernels), len(kernel_sizes))
        ks = [int(k) for k in kernel_sizes[:K]]
        arr = (C.c_int32 * len(ks))(*ks)
        vis = torch.empty((G, A, A), dtype=torch.bool, device=dev)
        shr = torch.empty((K, G, A, A), dtype=torch.bool, device=dev)
        nws = torch.empty((2, G, A, A), dtype=torch.uint8, device=dev)

        def visibility_levels():
            check(L.pdhip_texel_visibility_shapes(ptr(cp), V, S, ptr(gb), ptr(m8), A, ptr(uvc), ptr(uvs), padding, ptr(depth), R, 0.0001,
                                                  ptr(as_u8(vis)), stream()), 'pdhip_texel_visibility_shapes')
            check(L.pdhip_nbf_shrink_shapes(ptr(m8), ptr(as_u8(vis)), V, S, A, arr, K, ptr(as_u8(shr)), ptr(nws), stream()),
                  'pdhip_nbf_shrink_shapes')
        visibility_levels()
        # ---- P2b, P3 (per view: the one-shape entry points with S*V views), P3b
        hard_r = ou.resize_masks(hard, res) if cam_res != res else hard
        valid, pix = ou.get_point_validation_and_pixels(cam_res, puv, pdep, depth, res, offset=0.0001)
        if point_validation_by_o3d:
            vis_h = torch.empty((G, N), dtype=torch.bool, device=dev)
            hws = torch.empty((L.pdhip_hpr_ws_bytes(G, N),), dtype=torch.uint8, device=dev)
            check(L.pdhip_hidden_point_removal_shapes(ptr(coords), N, ptr(_eyes(camera_info['eye_positions'], S, dev)), V, S,
                                                      float(hidden_point_removal_radius), ptr(as_u8(valid)), ptr(as_u8(vis_h)), ptr(hws),
                                                      stream()), 'pdhip_hidden_point_removal_shapes')
            valid = vis_h
        # ---- P4-P6
        sparse = torch.empty((G, 3, res, res), device=dev)
        m0 = torch.empty_like(sparse)
        m2 = torch.empty_like(sparse)
        sf = torch.empty((G,), device=dev)
        sws = torch.empty((L.pdhip_sparse_views_ws_bytes(G, N, res),), dtype=torch.uint8, device=dev)
        check(L.pdhip_sparse_views_shapes(ptr(pix), ptr(colors), ptr(as_u8(valid)), ptr(as_u8(hard_r)), V, S, N, res, int(point_size),
                                          int(edge_point_size), float(mask_ratio_thresh), ptr(sparse), ptr(m0), ptr(m2), ptr(sf), None,
                                          ptr(sws), stream()), 'pdhip_sparse_views_shapes')
        # ---- I0 / I1 (all S*V views in one batch)
        inpainted = ou.get_inpainted_images(sparse, m0, m2, None, inpainter, G, method=texture_gen_method)
        # ---- (Uq1-Uq2, N1-N3 above,) Uq3-Uq4, Uq5
        atlas = torch.empty((S, A, A, 3), device=dev)
        painted = torch.empty((S, A, A), dtype=torch.bool, device=dev)
        view_ids = torch.empty((S, A, A), dtype=torch.int32, device=dev)
        bd = camera_info['base_dirs'].float().contiguous()
        check(L.pdhip_view_select_blend_shapes(ptr(cp), V, S, ptr(gb), ptr(m8), ptr(fid), A, ptr(fn), F, ptr(bd), ptr(uvc), ptr(uvs), padding,
                                               ptr(sf), ptr(as_u8(shr)), K, ptr(as_u8(vis)), 1, ptr(inpainted.float().contiguous()),
                                               int(res), ptr(atlas), ptr(as_u8(painted)), ptr(view_ids), stream()),
              'pdhip_view_select_blend_shapes')
        out = ou.nearest_fill(atlas, mask.reshape(S, A, A), 'HWC')          # unproject.dilate_atlas per shape, one launch set
    if return_intermediates:
        return dict(atlas=out, inpainted=inpainted, sparse=sparse, mask0=m0, mask2=m2, view_ids=view_ids, painted=painted,
                    shrinked=shr[K - 1], visibility=vis, point_validation=valid, scale_factors=sf, mesh_depths=depth)
    return out
