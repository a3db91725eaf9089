"""The texturing hot path as one call: project -> inpaint -> unproject (-> dilate).

`colorize_one_mesh` keeps the keyword surface of the reference's demo.colorize_one_mesh
(/root/reference/demo.py:38-253) for the stages this build covers.  Stages outside SURVEY 8 rows (a)-(e)
(`complete_unseen_by='optimize'`, `texture_gen_method='linear'`) raise NotImplementedError instead of silently doing
something else; `complete_unseen_by='unproject'`, `optimize_from=None` is the measured path.
"""
import torch

from . import ours_utils as ou
from . import unproject as up


def colorize_one_mesh(coords, colors, vertices, faces, f_normals, xatlas_dict, camera_info, view_num, res, cam_res,
                      device=None, save_img_path=None, point_validation_by_o3d=True,
                      refine_point_validation_by_remove_abnormal_depth=False, hidden_point_removal_radius=100,
                      texture_gen_method='DDNM_inpaint', point_size=1, edge_point_size=1, crop_img=True,
                      crop_padding=0.05, mask_ratio_thresh=0.82, optimize_from=None, edge_dilate_kernels=(21,),
                      complete_unseen_by='unproject', inpainter=None, glctx=None, logger=None,
                      xatlas_texture_res=1024, refine_res=512, return_intermediates=False, **kwargs):
    if refine_point_validation_by_remove_abnormal_depth:
        raise NotImplementedError("refine_point_validation_by_remove_abnormal_depth (off in every shipped config) is not built")
    if complete_unseen_by not in ('unproject', 'neighbor'):
        raise NotImplementedError(f"complete_unseen_by={complete_unseen_by!r}: 'unproject' and 'neighbor' are built ('optimize' "
                                  "needs the TextureField network, outside SURVEY 8)")
    if complete_unseen_by == 'neighbor' and (xatlas_dict.get('uvs') is None or xatlas_dict.get('mesh_tex_idx') is None):
        raise ValueError("complete_unseen_by='neighbor' needs xatlas_dict['uvs'] and ['mesh_tex_idx'] (demo.py:199-200)")
    if optimize_from not in (None, 'None', 'scratch', 'naive', 'ours'):
        raise ValueError(f"optimize_from={optimize_from!r}")
    cams = camera_info['cams']
    base_dirs = camera_info['base_dirs']
    eye_positions = camera_info['eye_positions']
    gb_pos = xatlas_dict['gb_pos']
    mask = xatlas_dict['mask']
    face_id = xatlas_dict['per_atlas_pixel_face_id']
    with torch.no_grad():
        hard_masks, face_idxs, mesh_depths, vertice_uvs, uv_centers, uv_scales, padding, point_uvs, point_depths = \
            ou.get_rendered_hard_mask_and_face_idx_batch(cams, vertices, faces, coords, glctx=glctx, rescale=crop_img,
                                                         padding=crop_padding)
        if cam_res != res:
            hard_masks = ou.resize_masks(hard_masks, res)
        point_validation, _ = ou.get_point_validation_by_depth(cam_res, point_uvs, point_depths, mesh_depths, offset=0.0001)
        if point_validation_by_o3d:
            from .hpr import hidden_point_removal
            # demo.py:108-110 ORs the two tests: only points the depth test rejected need the hull query
            point_validation = hidden_point_removal(coords, eye_positions, hidden_point_removal_radius,
                                                    already_valid=point_validation)
        point_pixels = ou.get_point_pixels(point_uvs, res)
        sparse_imgs, hard_mask0s, hard_mask2s, scale_factors = ou.get_sparse_images(
            point_pixels, colors, point_validation, hard_masks, save_img_path, view_num, res, point_size,
            edge_point_size, mask_ratio_thresh)
        inpainted = ou.get_inpainted_images(sparse_imgs, hard_mask0s, hard_mask2s, save_img_path, inpainter, view_num,
                                            method=texture_gen_method)
        atlas, shrinked, view_ids, painted, vis = up.unproject_dense(
            inpainted, f_normals, res, cams, cam_res, base_dirs, gb_pos, mask, face_id, uv_centers, uv_scales, padding,
            scale_factors, mesh_depths, list(edge_dilate_kernels), complete_unseen_by == 'unproject')
        if complete_unseen_by == 'neighbor':
            # demo.py:180-200: faces that still own unpainted texels -> subdivide, average over mesh neighbours, nearest fill
            tif = up.unpainted_face_ids(face_id, painted, faces.shape[0])
            atlas = up.paint_invisible_areas_by_neighbors(vertices, faces, xatlas_dict['uvs'], xatlas_dict['mesh_tex_idx'], tif,
                                                          atlas, painted, use_atlas=True)
        else:
            atlas = up.dilate_atlas(atlas, mask)
        if optimize_from not in (None, 'None'):
            # demo.py:211-236: 100 Adam steps of the atlas against the inpainted views (flip to image orientation and back)
            from .optimize import optimize_color
            init = None if optimize_from == 'scratch' else atlas.permute(2, 0, 1).flip(1).contiguous()
            shr = shrinked if optimize_from == 'ours' else None
            eyes_t = torch.tensor(eye_positions).float().to(atlas.device)
            opt, _ = optimize_color(init, inpainted, vertices, faces, xatlas_dict['uvs'], xatlas_dict['mesh_tex_idx'], cams, eyes_t,
                                    torch.zeros_like(eyes_t), camera_info.get('up_dirs'), uv_centers, uv_scales, padding,
                                    scale_factors, glctx, shrinked_per_view_per_pixel_visibility=shr)
            atlas = opt[0].flip(1).permute(1, 2, 0).contiguous()
    if return_intermediates:
        return dict(atlas=atlas, inpainted=inpainted, sparse=sparse_imgs, mask0=hard_mask0s, mask2=hard_mask2s,
                    view_ids=view_ids, painted=painted, shrinked=shrinked, visibility=vis,
                    point_validation=point_validation, scale_factors=scale_factors, mesh_depths=mesh_depths)
    return vertices, xatlas_dict.get('uvs'), faces, xatlas_dict.get('mesh_tex_idx'), atlas, mask


def colorize_meshes_batched(shapes, camera_info, view_num, res, cam_res, inpainter=None, texture_gen_method='DDNM_inpaint',
                            point_size=1, edge_point_size=1, crop_img=True, crop_padding=0.05, mask_ratio_thresh=0.82,
                            edge_dilate_kernels=(21,), point_validation_by_o3d=True, hidden_point_removal_radius=100, **unused):
    """Several independent shapes in one pass (BASELINE configs[4]: a batch of shapes per GPU): the projection / sparse-image
    stage runs per shape, the views of ALL shapes go through the inpainter together (one UNet batch of len(shapes) * V views --
    the 8x8 .. 32x32 levels of the UNet fill the chip better), the unprojection runs per shape.
    `shapes`: list of dicts with coords, colors, vertices, faces, f_normals, xatlas (gb_pos, mask, per_atlas_pixel_face_id).
    Same stages and results as colorize_one_mesh(complete_unseen_by='unproject', optimize_from=None); returns the atlases."""
    from .dist import _project_stage
    with torch.no_grad():
        prs = [_project_stage(sh['coords'], sh['colors'], sh['vertices'], sh['faces'], camera_info, view_num, res, cam_res, point_size,
                              edge_point_size, crop_img, crop_padding, mask_ratio_thresh, point_validation_by_o3d,
                              hidden_point_removal_radius) for sh in shapes]
        cat = lambda k: torch.cat([pr[k] for pr in prs], 0).contiguous()
        inpainted = ou.get_inpainted_images(cat('sparse'), cat('mask0'), cat('mask2'), None, inpainter, view_num * len(shapes),
                                            method=texture_gen_method)
        atlases = []
        for i, (sh, pr) in enumerate(zip(shapes, prs)):
            xat = sh['xatlas']
            atlas, _, _, _, _ = up.unproject_dense(
                inpainted[i * view_num:(i + 1) * view_num].contiguous(), sh['f_normals'], res, camera_info['cams'], cam_res,
                camera_info['base_dirs'], xat['gb_pos'], xat['mask'], xat['per_atlas_pixel_face_id'], pr['uv_centers'],
                pr['uv_scales'], pr['padding'], pr['scale_factors'], pr['mesh_depths'], list(edge_dilate_kernels), True)
            atlases.append(up.dilate_atlas(atlas, xat['mask']))
    return atlases
