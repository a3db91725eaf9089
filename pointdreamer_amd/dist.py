"""Multi-GPU sharding of the texturing path -- new functionality, the reference is single-GPU.

Two partitionings, one process per GPU (torch.distributed; backend 'nccl' == RCCL over xGMI on ROCm):
  * shape-parallel: independent shapes, no data-path collective (bench.py default, weak scaling; demo.files_of_rank);
  * view-parallel (this module): the V views of ONE shape are split across ranks.  A rank owns, for its views, everything that
    is per-view: P1-P6 (project / raster / visibility / sparse image), the DDNM inpainting (> 99 % of the time), Uq1-Uq2 (texel
    visibility) and N1-N3 (NBF shrink).  ONE all_gather then moves, per view, the inpainted image (3 r^2 f32), the raw and the
    shrunk texel visibility (A^2 BITS each level: 64-texel words) and four crop parameters; the cross-view part -- Uq3-Uq5 view selection,
    blend, dilation and the optional completion / optimisation stages -- runs replicated on every rank (34 P bytes of work),
    so every rank returns the full atlas and no second collective exists.
The stage functions are injectable so the sharding / gather logic is testable on CPU with gloo."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block partition of range(n_items): the first (n_items % world) ranks get one extra item."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def all_gather_views(local, n_views, rank, world, group=None, force_collective=False):
    """local [v_local, ...] (this rank's block of views, in shard_range order) -> [n_views, ...] on every rank.
    One all_gather; ragged blocks are padded to the largest block.  force_collective: run the collective at world size 1 too
    (a one-rank RCCL group: exercises the exact device-tensor path of the multi-GPU run on a single GPU)."""
    if world == 1 and not force_collective:
        return local
    vmax = (n_views + world - 1) // world
    pad = torch.zeros((vmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * vmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    pieces = [out[r * vmax:r * vmax + len(shard_range(n_views, r, world))] for r in range(world)]
    return torch.cat(pieces, 0)


def pack_bits(t):
    """Boolean / 0-1 byte DEVICE tensor [..., n] (n % 64 == 0) -> uint8 [..., n / 8], bit b of byte j = element 8 j + b (the 64-texel words of
    the NBF kernels as a byte stream): pdhip_pack_bits.  (The gloo tests of the gather logic run with injected host stages and inject their own
    host packer with them -- `stages['pack_bits']` / `stages['unpack_bits']`; nothing here computes on the CPU.)"""
    from . import _lib
    flat = t.reshape(-1).contiguous()
    n = flat.numel()
    if not flat.is_cuda:
        raise _lib.PdhipError("dist.pack_bits packs device tensors (pdhip_pack_bits); host-side callers inject stages['pack_bits']")
    assert n % 64 == 0, "bit-packed maps need a multiple of 64 texels"
    src = flat.view(torch.uint8) if flat.dtype == torch.bool else flat.to(torch.uint8)
    out = torch.empty((n // 8,), dtype=torch.uint8, device=flat.device)
    _lib.check(_lib.lib().pdhip_pack_bits(_lib.ptr(src), n, _lib.ptr(out), _lib.stream()), 'pdhip_pack_bits')
    return out.reshape(tuple(t.shape[:-1]) + (t.shape[-1] // 8,))


def unpack_bits(b, n):
    """Inverse of pack_bits: uint8 device tensor [..., n / 8] -> bool [..., n] (pdhip_unpack_bits)."""
    from . import _lib
    flat = b.reshape(-1).contiguous()
    if not flat.is_cuda:
        raise _lib.PdhipError("dist.unpack_bits unpacks device tensors (pdhip_unpack_bits); host-side callers inject stages['unpack_bits']")
    total = flat.numel() * 8
    out = torch.empty((total,), dtype=torch.uint8, device=flat.device)
    _lib.check(_lib.lib().pdhip_unpack_bits(_lib.ptr(flat), total, _lib.ptr(out), _lib.stream()), 'pdhip_unpack_bits')
    return out.view(torch.bool).reshape(tuple(b.shape[:-1]) + (n,))


def record_bytes(img_shape, A, K):
    """Bytes of one view's record: image f32 + (1 + K) bit-packed A x A maps + four f32 crop parameters."""
    return 4 * img_shape[0] * img_shape[1] * img_shape[2] + (1 + K) * (A * A // 8) + 16


def pack_view_records(inpainted, vis, per_kernel, uv_centers, uv_scales, scale_factors, pack=None):
    """One byte record per view: [image f32 | visibility, 1 bit per texel | K shrunk levels, 1 bit per texel | centre x, centre y, scale,
    factor f32] -- SURVEY 8(e)'s payload: 786 KB + (1 + K) x 128 KiB per view at r = 256, A = 1024 (round 5 shipped the maps as bytes: 2.8 MB)."""
    pack = pack or pack_bits
    v = inpainted.shape[0]
    dev = inpainted.device
    A2 = vis.shape[-1] * vis.shape[-2]
    par = torch.cat([uv_centers.reshape(v, 2).float(), uv_scales.reshape(v, 1).float(), scale_factors.reshape(v, 1).float()], 1)
    parts = [inpainted.reshape(v, -1).float().contiguous().view(torch.uint8),
             pack(vis.reshape(v, A2)),
             pack(per_kernel.permute(1, 0, 2, 3).reshape(v, -1, A2)).reshape(v, -1),
             par.contiguous().view(torch.uint8)]
    return torch.cat([p.to(dev) for p in parts], 1).contiguous()


def unpack_view_records(rec, img_shape, A, K, unpack=None):
    """Inverse of pack_view_records for all V views: (inpainted [V,3,r,r], vis [V,A,A] bool, per_kernel [K,V,A,A] bool,
    uv_centers [V,1,2], uv_scales [V,1,1], scale_factors [V])."""
    unpack = unpack or unpack_bits
    V = rec.shape[0]
    n_img = 4 * img_shape[0] * img_shape[1] * img_shape[2]
    nb = A * A // 8
    o = 0
    img = rec[:, o:o + n_img].contiguous().view(torch.float32).reshape(V, *img_shape); o += n_img
    vis = unpack(rec[:, o:o + nb].contiguous(), A * A).reshape(V, A, A); o += nb
    pk = unpack(rec[:, o:o + K * nb].contiguous().reshape(V, K, nb), A * A).reshape(V, K, A, A).permute(1, 0, 2, 3).contiguous(); o += K * nb
    par = rec[:, o:o + 16].contiguous().view(torch.float32).reshape(V, 4)
    return img, vis, pk, par[:, 0:2].reshape(V, 1, 2).contiguous(), par[:, 2].reshape(V, 1, 1).contiguous(), par[:, 3].contiguous()


def _subset_camera_info(camera_info, sl):
    out = dict(camera_info)
    for k in ('cams', 'base_dirs', 'eye_positions', 'up_dirs', 'cam_RTs'):
        if camera_info.get(k) is not None:
            out[k] = camera_info[k][sl]
    return out


def default_stages():
    """The real (HIP) stages: pipeline._before_inpaint / ours_utils.get_inpainted_images / unproject.per_view_visibility /
    pipeline._after_inpaint."""
    from . import ours_utils as ou, unproject as up, pipeline as pl

    def before(coords, colors, vertices, faces, cam_info_local, n_local, res, cam_res, save_img_path, opts, view_offset):
        return pl._before_inpaint(coords, colors, vertices, faces, cam_info_local, n_local, res, cam_res, save_img_path,
                                  opts['point_validation_by_o3d'], opts['hidden_point_removal_radius'], opts['point_size'],
                                  opts['edge_point_size'], opts['crop_img'], opts['crop_padding'], opts['mask_ratio_thresh'],
                                  view_offset=view_offset, refine_point_validation=opts.get('refine_point_validation', False),
                                  refine_res=opts.get('refine_res', 512))

    def inpaint(pre, save_img_path, inpainter, n_local, method, first_key, advance, view_offset):
        return ou.get_inpainted_images(pre['sparse'], pre['mask0'], pre['mask2'], save_img_path, inpainter, n_local, method=method,
                                       first_key=first_key, advance=advance, view_offset=view_offset)

    def visibility(pre, cam_info_local, cam_res, xatlas_dict, edge_dilate_kernels, save_img_path, view_offset):
        import os
        sp = None if save_img_path is None else os.path.join(save_img_path, 'shrink_per_view_edge')
        return up.per_view_visibility(cam_info_local['cams'], cam_res, xatlas_dict['gb_pos'], xatlas_dict['mask'], pre['uv_centers'],
                                      pre['uv_scales'], pre['padding'], pre['mesh_depths'], edge_dilate_kernels, sp, view_offset)

    def after(pre_all, inpainted, vis, per_kernel, vertices, faces, f_normals, xatlas_dict, camera_info, res, cam_res,
              edge_dilate_kernels, complete_unseen_by, optimize_from):
        atlas, _ = pl._after_inpaint(pre_all, inpainted, vertices, faces, f_normals, xatlas_dict, camera_info, res, cam_res,
                                     edge_dilate_kernels, complete_unseen_by, optimize_from, vis_and_shrunk=(vis, per_kernel))
        return atlas
    return dict(before=before, inpaint=inpaint, visibility=visibility, after=after)


def colorize_one_mesh_view_parallel(coords, colors, vertices, faces, f_normals, xatlas_dict, camera_info, view_num, res,
                                    cam_res, rank, world, inpainter=None, texture_gen_method='DDNM_inpaint',
                                    point_size=1, edge_point_size=1, crop_img=True, crop_padding=0.05,
                                    mask_ratio_thresh=0.82, edge_dilate_kernels=(21,), point_validation_by_o3d=True,
                                    hidden_point_removal_radius=100, complete_unseen_by='unproject', optimize_from=None,
                                    save_img_path=None, group=None, stages=None, shape_key=None, return_full=False,
                                    force_collective=False, refine_point_validation_by_remove_abnormal_depth=False, refine_res=512,
                                    **unused):
    """View-parallel demo.colorize_one_mesh: same keyword surface and defaults as pipeline.colorize_one_mesh, same atlas on
    every rank (bit-identical to the single-process result for the index / copy stages; for DDNM the noise of view k is keyed by
    its global index, so the result does not depend on `world` either).  shape_key: running index of the shape (noise key base
    = shape_key * view_num); default = the inpainter's own image counter, which every rank advances by view_num."""
    from . import pipeline as pl
    pl._check_options(xatlas_dict, refine_point_validation_by_remove_abnormal_depth, complete_unseen_by, optimize_from)
    if world > view_num:
        # a rank without views would enter the per-view stages with V = 0 (the HIP entry points require V > 0) while the others
        # block in the all_gather: refuse up front, on every rank alike
        raise ValueError(f"view-parallel needs world size <= view_num (world {world}, view_num {view_num}): "
                         f"use shape-parallel for the remaining GPUs")
    st = default_stages() if stages is None else stages
    mine = shard_range(view_num, rank, world)
    sl = slice(mine.start, mine.stop)
    opts = dict(point_validation_by_o3d=point_validation_by_o3d, hidden_point_removal_radius=hidden_point_removal_radius,
                point_size=point_size, edge_point_size=edge_point_size, crop_img=crop_img, crop_padding=crop_padding,
                mask_ratio_thresh=mask_ratio_thresh, refine_point_validation=refine_point_validation_by_remove_abnormal_depth,
                refine_res=refine_res)
    A = xatlas_dict['mask'].shape[1]
    with torch.no_grad():
        cam_local = _subset_camera_info(camera_info, sl)
        pre = st['before'](coords, colors, vertices, faces, cam_local, len(mine), res, cam_res, save_img_path, opts, mine.start)
        base = (inpainter._images if (shape_key is None and inpainter is not None) else (shape_key or 0) * view_num)
        local = st['inpaint'](pre, save_img_path, inpainter, len(mine), texture_gen_method, base + mine.start, view_num, mine.start)
        vis_l, pk_l = st['visibility'](pre, cam_local, cam_res, xatlas_dict, list(edge_dilate_kernels), save_img_path, mine.start)
        K = pk_l.shape[0]
        dev = local.device
        v = len(mine)
        uvc = pre['uv_centers'] if torch.is_tensor(pre['uv_centers']) else torch.full((v, 1, 2), float(pre['uv_centers'] or 0.0), device=dev)
        uvs = pre['uv_scales'] if torch.is_tensor(pre['uv_scales']) else torch.full((v, 1, 1), float(pre['uv_scales'] or 2.0), device=dev)
        sf = pre['scale_factors'] if torch.is_tensor(pre['scale_factors']) else torch.ones((v,), device=dev)
        rec = pack_view_records(local, vis_l, pk_l, uvc, uvs, sf, pack=st.get('pack_bits'))
        rec = all_gather_views(rec, view_num, rank, world, group, **({'force_collective': True} if force_collective else {}))  # the one collective
        inpainted, vis, per_kernel, uvc_a, uvs_a, sf_a = unpack_view_records(rec, tuple(local.shape[1:]), A, K, unpack=st.get('unpack_bits'))
        pre_all = dict(uv_centers=uvc_a, uv_scales=uvs_a, padding=pre['padding'], scale_factors=sf_a, mesh_depths=None)
        atlas = st['after'](pre_all, inpainted, vis, per_kernel, vertices, faces, f_normals, xatlas_dict, camera_info, res, cam_res,
                            edge_dilate_kernels, complete_unseen_by, optimize_from)
    if return_full:
        return vertices, xatlas_dict.get('uvs'), faces, xatlas_dict.get('mesh_tex_idx'), atlas, xatlas_dict['mask']
    return atlas
