"""The texturing hot path as one call: project -> inpaint -> unproject (-> dilate).

`colorize_one_mesh` keeps the keyword surface of the reference's demo.colorize_one_mesh
(/root/reference/demo.py:38-253).  Built: texture_gen_method 'nearest' / 'linear' / 'DDNM_inpaint', complete_unseen_by
'unproject' / 'neighbor', optimize_from None / 'scratch' / 'naive' / 'ours', `refine_point_validation_by_remove_abnormal_depth`
(ours_utils.refine_point_validation; off in every shipped config).  Not built, and refused with NotImplementedError instead of
silently doing something else: `complete_unseen_by='optimize'` (a per-shape TextureField training loop, outside SURVEY 8).  The
measured path of bench.py is `complete_unseen_by='unproject'`, `optimize_from=None`.
"""
import torch

from . import ours_utils as ou
from . import unproject as up


def _check_options(xatlas_dict, refine_point_validation_by_remove_abnormal_depth, complete_unseen_by, optimize_from):
    if complete_unseen_by not in ('unproject', 'neighbor'):
        raise NotImplementedError(f"complete_unseen_by={complete_unseen_by!r}: 'unproject' and 'neighbor' are built ('optimize' "
                                  "needs the TextureField network, outside SURVEY 8)")
    if complete_unseen_by == 'neighbor' and (xatlas_dict.get('uvs') is None or xatlas_dict.get('mesh_tex_idx') is None):
        raise ValueError("complete_unseen_by='neighbor' needs xatlas_dict['uvs'] and ['mesh_tex_idx'] (demo.py:199-200)")
    if optimize_from not in (None, 'None', 'scratch', 'naive', 'ours'):
        raise ValueError(f"optimize_from={optimize_from!r}")


def _before_inpaint(coords, colors, vertices, faces, camera_info, view_num, res, cam_res, save_img_path, point_validation_by_o3d,
                    hidden_point_removal_radius, point_size, edge_point_size, crop_img, crop_padding, mask_ratio_thresh, glctx=None,
                    view_offset=0, refine_point_validation=False, refine_res=512):
    """demo.py:93-129: project, rasterise, visibility, sparse views -- everything of one shape ahead of the inpainter.
    Every view is independent here: `camera_info` may hold a subset of a shape's cameras (view-parallel sharding), view_offset
    is then the index of its first view in the per-view file names."""
    cams = camera_info['cams']
    hard_masks, face_idxs, mesh_depths, vertice_uvs, uv_centers, uv_scales, padding, point_uvs, point_depths = \
        ou.get_rendered_hard_mask_and_face_idx_batch(cams, vertices, faces, coords, glctx=glctx, rescale=crop_img,
                                                     padding=crop_padding)
    if cam_res != res:
        hard_masks = ou.resize_masks(hard_masks, res)
    point_validation, point_pixels = ou.get_point_validation_and_pixels(cam_res, point_uvs, point_depths, mesh_depths, res, offset=0.0001)
    if point_validation_by_o3d:
        from .hpr import hidden_point_removal
        # demo.py:108-110 ORs the two tests: only points the depth test rejected need the hull query
        point_validation = hidden_point_removal(coords, camera_info['eye_positions'], hidden_point_removal_radius,
                                                already_valid=point_validation)
    if refine_point_validation:                              # demo.py:115-117 (False by default)
        cam_RTs = camera_info.get('cam_RTs')
        if cam_RTs is None:                                  # demo.py:334-335
            from .camera_utils import get_cam_Ks_RTs_from_locations
            cam_RTs = get_cam_Ks_RTs_from_locations(camera_info['eye_positions'])[1]
        point_validation = ou.refine_point_validation(cam_RTs, camera_info.get('cam_K'), refine_res, hard_masks, point_validation,
                                                      point_uvs, coords, save_img_path, view_offset=view_offset)
    sparse_imgs, hard_mask0s, hard_mask2s, scale_factors = ou.get_sparse_images(
        point_pixels, colors, point_validation, hard_masks, save_img_path, view_num, res, point_size,
        edge_point_size, mask_ratio_thresh, view_offset=view_offset)
    return dict(sparse=sparse_imgs, mask0=hard_mask0s, mask2=hard_mask2s, scale_factors=scale_factors, uv_centers=uv_centers,
                uv_scales=uv_scales, padding=padding, mesh_depths=mesh_depths, point_validation=point_validation)


def _after_inpaint(pre, inpainted, vertices, faces, f_normals, xatlas_dict, camera_info, res, cam_res, edge_dilate_kernels,
                   complete_unseen_by, optimize_from, glctx=None, save_img_path=None, vis_and_shrunk=None):
    """demo.py:167-236: unproject, complete the unseen texels, optionally optimise.  Returns (atlas, intermediates).
    vis_and_shrunk: per-view visibility + NBF levels gathered from the ranks that own the views (dist.py); else computed here."""
    cams, base_dirs, eye_positions = camera_info['cams'], camera_info['base_dirs'], camera_info['eye_positions']
    gb_pos, mask, face_id = xatlas_dict['gb_pos'], xatlas_dict['mask'], xatlas_dict['per_atlas_pixel_face_id']
    atlas, shrinked, view_ids, painted, vis = up.unproject_dense(
        inpainted, f_normals, res, cams, cam_res, base_dirs, gb_pos, mask, face_id, pre['uv_centers'], pre['uv_scales'],
        pre['padding'], pre['scale_factors'], pre.get('mesh_depths'), list(edge_dilate_kernels), complete_unseen_by == 'unproject',
        save_img_path=save_img_path, vis_and_shrunk=vis_and_shrunk)
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
                                torch.zeros_like(eyes_t), camera_info.get('up_dirs'), pre['uv_centers'], pre['uv_scales'],
                                pre['padding'], pre['scale_factors'], glctx, shrinked_per_view_per_pixel_visibility=shr)
        atlas = opt[0].flip(1).permute(1, 2, 0).contiguous()
    return atlas, dict(view_ids=view_ids, painted=painted, shrinked=shrinked, visibility=vis)


def colorize_one_mesh(coords, colors, vertices, faces, f_normals, xatlas_dict, camera_info, view_num, res, cam_res,
                      device=None, save_img_path=None, point_validation_by_o3d=True,
                      refine_point_validation_by_remove_abnormal_depth=False, hidden_point_removal_radius=100,
                      texture_gen_method='DDNM_inpaint', point_size=1, edge_point_size=1, crop_img=True,
                      crop_padding=0.05, mask_ratio_thresh=0.82, optimize_from=None, edge_dilate_kernels=(21,),
                      complete_unseen_by='unproject', inpainter=None, glctx=None, logger=None,
                      xatlas_texture_res=1024, refine_res=512, return_intermediates=False, reuse_inpainted=True, **kwargs):
    _check_options(xatlas_dict, refine_point_validation_by_remove_abnormal_depth, complete_unseen_by, optimize_from)
    with torch.no_grad():
        pre = _before_inpaint(coords, colors, vertices, faces, camera_info, view_num, res, cam_res, save_img_path,
                              point_validation_by_o3d, hidden_point_removal_radius, point_size, edge_point_size, crop_img,
                              crop_padding, mask_ratio_thresh, glctx,
                              refine_point_validation=refine_point_validation_by_remove_abnormal_depth, refine_res=refine_res)
        # demo.py:138-147: every {i}_inpainted.png already on disk -> load them instead of inpainting again (resume surface)
        inpainted = ou.load_inpainted_images(save_img_path, view_num, coords.device) if reuse_inpainted else None
        if inpainted is None:
            inpainted = ou.get_inpainted_images(pre['sparse'], pre['mask0'], pre['mask2'], save_img_path, inpainter, view_num,
                                                method=texture_gen_method)
        elif logger is not None:
            logger.info('All inpainted images exist, load them instead of inpainting again')
        atlas, post = _after_inpaint(pre, inpainted, vertices, faces, f_normals, xatlas_dict, camera_info, res, cam_res,
                                     edge_dilate_kernels, complete_unseen_by, optimize_from, glctx, save_img_path)
    if return_intermediates:
        return dict(atlas=atlas, inpainted=inpainted, sparse=pre['sparse'], mask0=pre['mask0'], mask2=pre['mask2'],
                    view_ids=post['view_ids'], painted=post['painted'], shrinked=post['shrinked'], visibility=post['visibility'],
                    point_validation=pre['point_validation'], scale_factors=pre['scale_factors'], mesh_depths=pre['mesh_depths'])
    return vertices, xatlas_dict.get('uvs'), faces, xatlas_dict.get('mesh_tex_idx'), atlas, xatlas_dict['mask']


_STREAMS = {}


def _shape_streams(dev, n):
    key = str(dev)
    pool = _STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


def colorize_meshes_batched(shapes, camera_info, view_num, res, cam_res, inpainter=None, texture_gen_method='DDNM_inpaint',
                            point_size=1, edge_point_size=1, crop_img=True, crop_padding=0.05, mask_ratio_thresh=0.82,
                            edge_dilate_kernels=(21,), point_validation_by_o3d=True, hidden_point_removal_radius=100,
                            complete_unseen_by='unproject', optimize_from=None, save_img_paths=None, return_full=False,
                            refine_point_validation_by_remove_abnormal_depth=False, concurrent=True, **unused):
    """Several independent shapes in one pass (BASELINE configs[4]: a batch of shapes per GPU): the projection / sparse-image
    stage runs per shape, the views of ALL shapes go through the inpainter together (one UNet batch of len(shapes) * V views --
    the 8x8 .. 32x32 levels of the UNet fill the chip better), unprojection / completion / optimisation run per shape.
    `shapes`: list of dicts with coords, colors, vertices, faces, f_normals, xatlas (gb_pos, mask, per_atlas_pixel_face_id
    [, uvs, mesh_tex_idx]).  Same stages and results as colorize_one_mesh shape by shape -- also for DDNM: the noise of a view is
    keyed by the running image count of the inpainter (Inpainter.inpaint_views), which advances by V per shape either way;
    `save_img_paths`: one directory per shape for the per-view PNGs.  Returns the atlases, or colorize_one_mesh's 6-tuples with
    return_full.  (The `{i}_inpainted.png` reuse of demo.py:138-147 applies per shape only in colorize_one_mesh.)"""
    for sh in shapes:
        _check_options(sh['xatlas'], refine_point_validation_by_remove_abnormal_depth, complete_unseen_by, optimize_from)
    paths = list(save_img_paths) if save_img_paths is not None else [None] * len(shapes)
    # shapes of equal sizes, plain options: ONE launch per stage for all of them (pointdreamer_amd/shapes.py, pdhip_*_shapes)
    from . import shapes as _shp
    if (unused.get('stacked_launches', True) and len(shapes) > 1 and len(shapes) * view_num <= 64 and complete_unseen_by == 'unproject' and
            optimize_from in (None, 'None') and not refine_point_validation_by_remove_abnormal_depth and all(pth is None for pth in paths) and
            texture_gen_method in ('nearest', 'linear', 'DDNM_inpaint') and shapes[0]['coords'].is_cuda and _shp.uniform(shapes)):
        atl = _shp.colorize_shapes(_shp.stack(shapes), camera_info, view_num, res, cam_res, inpainter=inpainter,
                                   texture_gen_method=texture_gen_method, point_size=point_size, edge_point_size=edge_point_size,
                                   crop_img=crop_img, crop_padding=crop_padding, mask_ratio_thresh=mask_ratio_thresh,
                                   edge_dilate_kernels=edge_dilate_kernels, point_validation_by_o3d=point_validation_by_o3d,
                                   hidden_point_removal_radius=hidden_point_removal_radius)
        outs = []
        for i, sh in enumerate(shapes):
            xat = sh['xatlas']
            outs.append((sh['vertices'], xat.get('uvs'), sh['faces'], xat.get('mesh_tex_idx'), atl[i], xat['mask']) if return_full else atl[i])
        return outs
    # The per-shape stages of different shapes are independent and mostly latency-bound (a few hundred wavefronts per kernel): each
    # shape's stages are queued on a HIP stream of its own so that the GPU overlaps them; the inpainter runs on the caller's stream.
    dev = shapes[0]['coords'].device
    use_streams = concurrent and len(shapes) > 1 and dev.type == 'cuda'
    main = torch.cuda.current_stream(dev) if use_streams else None
    streams = _shape_streams(dev, len(shapes)) if use_streams else [None] * len(shapes)

    def on_stream(st, fn):
        if st is None:
            return fn()
        st.wait_stream(main)
        with torch.cuda.stream(st):
            return fn()

    def back_to_main(st, value):              # results of a side stream: the caller's stream waits, the allocator learns the new user
        if st is None:
            return
        main.wait_stream(st)
        for t in (value.values() if isinstance(value, dict) else (value if isinstance(value, (tuple, list)) else (value,))):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)

    with torch.no_grad():
        pres = [on_stream(st, lambda sh=sh, pth=pth: _before_inpaint(
                    sh['coords'], sh['colors'], sh['vertices'], sh['faces'], camera_info, view_num, res, cam_res, pth,
                    point_validation_by_o3d, hidden_point_removal_radius, point_size, edge_point_size, crop_img, crop_padding,
                    mask_ratio_thresh, refine_point_validation=refine_point_validation_by_remove_abnormal_depth,
                    refine_res=unused.get('refine_res', 512))) for sh, pth, st in zip(shapes, paths, streams)]
        for st, pr in zip(streams, pres):
            back_to_main(st, pr)
        cat = lambda k: torch.cat([pr[k] for pr in pres], 0).contiguous()
        inpainted = ou.get_inpainted_images(cat('sparse'), cat('mask0'), cat('mask2'), None, inpainter, view_num * len(shapes),
                                            method=texture_gen_method)
        outs, atlases = [], []
        for i, (sh, pr, pth, st) in enumerate(zip(shapes, pres, paths, streams)):
            inp = inpainted[i * view_num:(i + 1) * view_num].contiguous()
            if pth is not None:
                ou.save_inpainted_images(inp, pr['mask0'], pth, view_num, texture_gen_method)
            atlas = on_stream(st, lambda sh=sh, pr=pr, pth=pth, inp=inp: _after_inpaint(
                pr, inp, sh['vertices'], sh['faces'], sh['f_normals'], sh['xatlas'], camera_info, res, cam_res, edge_dilate_kernels,
                complete_unseen_by, optimize_from, None, pth)[0])
            if st is not None:
                inp.record_stream(st)
                for t in pr.values():
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(st)
            atlases.append(atlas)
        for st, atlas in zip(streams, atlases):
            back_to_main(st, atlas)
        for sh, atlas in zip(shapes, atlases):
            xat = sh['xatlas']
            outs.append((sh['vertices'], xat.get('uvs'), sh['faces'], xat.get('mesh_tex_idx'), atlas, xat['mask']) if return_full
                        else atlas)
    return outs


class ShapeGraphs:
    """The texturing path of one shape captured into HIP graphs (`torch.cuda.CUDAGraph`), `n_slots` of them on streams of their own:
    a shape's ~45 launches (0.26 ms of host enqueue, more than half of its 0.54 ms) become one graph launch, and the latency-bound
    kernels of different shapes overlap on the GPU -- 0.31 ms per shape at 8 slots against 0.45 ms for eager launches on streams
    (30k-point clouds, 8 x 256^2 views, hidden-point removal on).  Results equal the eager path bit for bit (tested).

    Only sync-free configurations can be captured: texture_gen_method='nearest', complete_unseen_by='unproject', optimize_from=None,
    no per-view files.  The cloud size, the mesh, the atlas and the cameras are fixed at capture; `run` takes the clouds."""

    def __init__(self, n_slots, n_points, vertices, faces, f_normals, xatlas_dict, camera_info, view_num, res, cam_res, **cfg):
        if cfg.get('texture_gen_method', 'nearest') != 'nearest' or cfg.get('complete_unseen_by', 'unproject') != 'unproject' or \
                cfg.get('optimize_from') not in (None, 'None') or cfg.get('save_img_path') is not None:
            raise ValueError("ShapeGraphs captures the sync-free configuration only: texture_gen_method='nearest', "
                             "complete_unseen_by='unproject', optimize_from=None, save_img_path=None")
        cfg = dict(cfg, texture_gen_method='nearest', complete_unseen_by='unproject', optimize_from=None, inpainter=None,
                   save_img_path=None)
        dev = vertices.device
        self.n_points = int(n_points)
        self._slots = []
        run_one = lambda p, c: colorize_one_mesh(p, c, vertices, faces, f_normals, xatlas_dict, camera_info, view_num, res, cam_res,
                                                 **cfg)[4]
        for _ in range(int(n_slots)):
            st = torch.cuda.Stream(device=dev)
            pts = torch.zeros((self.n_points, 3), device=dev)
            col = torch.zeros((self.n_points, 3), device=dev)
            pts[:, 2] = 1.0                                                   # (any cloud: the warm-up runs fill the host-side caches)
            gr = torch.cuda.CUDAGraph()
            st.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st):
                for _ in range(2):
                    run_one(pts, col)
                st.synchronize()
                with torch.cuda.graph(gr, stream=st):
                    out = run_one(pts, col)
            self._slots.append((st, gr, pts, col, out))
        torch.cuda.current_stream(dev).wait_stream(self._slots[-1][0])

    def __len__(self):
        return len(self._slots)

    def run(self, clouds, clone=True):
        """clouds: up to n_slots pairs (coords [n_points,3], colors [n_points,3]) -> their atlases [A,A,3] (in slot order).
        clone=False returns the slots' own output tensors, valid until the slot runs again."""
        if len(clouds) > len(self._slots):
            raise ValueError(f"{len(clouds)} clouds for {len(self._slots)} graph slots")
        main = torch.cuda.current_stream(self._slots[0][2].device)
        outs = []
        for (st, gr, pts, col, out), (p, c) in zip(self._slots, clouds):
            if tuple(p.shape) != (self.n_points, 3):
                raise ValueError(f"the graphs were captured for clouds of {self.n_points} points, got {tuple(p.shape)}")
            st.wait_stream(main)
            with torch.cuda.stream(st):
                pts.copy_(p, non_blocking=True); col.copy_(c, non_blocking=True)
                gr.replay()
                outs.append(out.clone() if clone else out)
        for st, *_ in self._slots[:len(clouds)]:
            main.wait_stream(st)
        for o in outs:
            o.record_stream(main)
        return outs
