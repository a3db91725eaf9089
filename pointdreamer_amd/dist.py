"""Multi-GPU sharding of the texturing path -- new functionality, the reference is single-GPU.

Two partitionings, one process per GPU (torch.distributed; backend 'nccl' == RCCL over xGMI on ROCm):
  * shape-parallel: independent shapes, no data-path collective (bench.py default, weak scaling);
  * view-parallel (this module): the V views of ONE shape are split across ranks for the inpainting stage --
    >99 % of a DDNM shape -- and assembled with a single all_gather of the inpainted images
    (V/world x 3 x r x r f32 = 786 KB per view); projection (sub-millisecond) and the NBF unprojection
    (34 P bytes of work) run replicated on every rank so no second collective is needed.
The stage functions are injectable so the sharding / gather logic is testable on CPU with gloo."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block partition of range(n_items): the first (n_items % world) ranks get one extra item."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def all_gather_views(local, n_views, rank, world, group=None):
    """local [v_local, ...] (this rank's block of views, in shard_range order) -> [n_views, ...] on every rank.
    One all_gather; ragged blocks are padded to the largest block."""
    if world == 1:
        return local
    vmax = (n_views + world - 1) // world
    pad = torch.zeros((vmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * vmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    pieces = [out[r * vmax:r * vmax + len(shard_range(n_views, r, world))] for r in range(world)]
    return torch.cat(pieces, 0)


def colorize_one_mesh_view_parallel(coords, colors, vertices, faces, f_normals, xatlas_dict, camera_info, view_num, res,
                                    cam_res, rank, world, inpainter=None, texture_gen_method='DDNM_inpaint',
                                    point_size=1, edge_point_size=1, crop_img=True, crop_padding=0.05,
                                    mask_ratio_thresh=0.82, edge_dilate_kernels=(21,), point_validation_by_o3d=False,
                                    hidden_point_removal_radius=100, group=None, stages=None, **unused):
    """View-parallel demo.colorize_one_mesh (complete_unseen_by='unproject', optimize_from=None).
    Every rank returns the full atlas [A,A,3]."""
    if stages is None:
        from . import ours_utils as ou, unproject as up
        stages = dict(project=_project_stage, inpaint=ou.get_inpainted_images, unproject=up.unproject_dense,
                      dilate=up.dilate_atlas)
    mine = shard_range(view_num, rank, world)
    with torch.no_grad():
        pr = stages['project'](coords, colors, vertices, faces, camera_info, view_num, res, cam_res, point_size,
                               edge_point_size, crop_img, crop_padding, mask_ratio_thresh, point_validation_by_o3d,
                               hidden_point_removal_radius)
        sl = slice(mine.start, mine.stop)
        local = stages['inpaint'](pr['sparse'][sl].contiguous(), pr['mask0'][sl].contiguous(), pr['mask2'][sl].contiguous(),
                                  None, inpainter, len(mine), method=texture_gen_method)
        inpainted = all_gather_views(local, view_num, rank, world, group)          # the one collective
        atlas, shr, view_ids, painted, vis = stages['unproject'](
            inpainted, f_normals, res, camera_info['cams'], cam_res, camera_info['base_dirs'], xatlas_dict['gb_pos'],
            xatlas_dict['mask'], xatlas_dict['per_atlas_pixel_face_id'], pr['uv_centers'], pr['uv_scales'], pr['padding'],
            pr['scale_factors'], pr['mesh_depths'], list(edge_dilate_kernels), True)
        atlas = stages['dilate'](atlas, xatlas_dict['mask'])
    return atlas


def _project_stage(coords, colors, vertices, faces, camera_info, view_num, res, cam_res, point_size, edge_point_size,
                   crop_img, crop_padding, mask_ratio_thresh, point_validation_by_o3d, hpr_radius):
    from . import ours_utils as ou
    cams = camera_info['cams']
    hard, fidx, depth, vuv, uvc, uvs, pad, puv, pdep = ou.get_rendered_hard_mask_and_face_idx_batch(
        cams, vertices, faces, coords, rescale=crop_img, padding=crop_padding)
    hard_r = ou.resize_masks(hard, res) if cam_res != res else hard
    pv, _ = ou.get_point_validation_by_depth(cam_res, puv, pdep, depth, offset=0.0001)
    if point_validation_by_o3d:
        from .hpr import hidden_point_removal
        pv = hidden_point_removal(coords, camera_info['eye_positions'], hpr_radius, already_valid=pv)
    pp = ou.get_point_pixels(puv, res)
    sparse, m0, m2, sf = ou.get_sparse_images(pp, colors, pv, hard_r, None, view_num, res, point_size, edge_point_size,
                                              mask_ratio_thresh)
    return dict(sparse=sparse, mask0=m0, mask2=m2, scale_factors=sf, uv_centers=uvc, uv_scales=uvs, padding=pad,
                mesh_depths=depth)
