"""demo.py-compatible driver for the texturing path: same CLI (`--config`, `--pc_file` file-or-directory),
same YAML keys (configs/*.yaml of the reference) and the same output tree as /root/reference/demo.py:311-497:

  <output_path>/<ply stem>_<config stem>/config.yaml, input_pc.ply,
      models/model_normalized.{obj,mtl,png}, others/{k}_{sparse,mask0,mask2,inpainted}.png, others/atlas_wo_background.png

Geometry and UV unwrapping are UPSTREAM of this build (POCO / SPR / xatlas are out of scope):
the driver uses the reference's own drop-in hooks -- `<pc>_untextured_mesh.obj` next to the PLY (demo.py:391-399)
and the cached `geo/xatlas_<res>.pth` dict (demo.py:428-448); if no mesh is supplied it falls back to the build's
stand-in UV sphere fitted to the cloud (clearly logged), so that the texturing path can be exercised end to end.

  python -m pointdreamer_amd.demo --config configs/nearest.yaml --pc_file dataset/demo_data/clock.ply
"""
import argparse
import datetime
import logging
import os
import shutil
import time

import numpy as np
import PIL.Image
import torch
import yaml

from . import io_utils, pipeline, synthetic
from .camera_utils import create_cameras

SUPPORTED_KEYS = ('texture_gen_method', 'camera_distribution', 'cam_res', 'view_num', 'res', 'point_size', 'edge_point_size',
                  'point_validation_by_o3d', 'hidden_point_removal_radius', 'refine_point_validation_by_remove_abnormal_depth',
                  'crop_img', 'crop_padding', 'mask_ratio_thresh', 'edge_dilate_kernels', 'optimize_from',
                  'xatlas_texture_res', 'complete_unseen_by', 'output_path')


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def get_logger(path):
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    logger = logging.getLogger('pointdreamer_amd')
    logger.setLevel(logging.INFO)
    if not logger.handlers:
        fmt = logging.Formatter('%(asctime)s %(levelname)s %(message)s')
        for h in (logging.FileHandler(path), logging.StreamHandler()):
            h.setFormatter(fmt)
            logger.addHandler(h)
    return logger


# keys of the reference's configs/*.yaml that steer stages UPSTREAM or outside of the texturing path (dataset bookkeeping, POCO / SPR
# geometry, evaluation): accepted verbatim and ignored, so that the reference's five YAML files load unchanged
UPSTREAM_KEYS = ('exp_name', 'exist_root_path', 'dataset_name', 'cls_id', 'input_pc_generate_method', 'demo', 'geo_root', 'geo_from',
                 'load_exist_dense_img_path', 'use_GT_geo_watertight', 'use_GT_multi_view_img', 'noise_stddev', 'coords_scale',
                 'input_type', 'input_already_noisy', 'save_dir', 'render_after_inference', 'save_input_pc', 'project2mesh',
                 'refine_res', 'smooth_mesh', 'sample_num')
DEFAULTS = dict(camera_distribution='fibonacci_sphere', cam_res=512, view_num=8, res=256, point_size=1, edge_point_size=1,
                point_validation_by_o3d=True, hidden_point_removal_radius=100, refine_point_validation_by_remove_abnormal_depth=False,
                crop_img=True, crop_padding=0.05, mask_ratio_thresh=0.82, edge_dilate_kernels=[21], optimize_from='ours',
                xatlas_texture_res=1024, complete_unseen_by='neighbor', output_path='output')


def load_config(cfg_file, overrides=None):
    """demo.py:313-314 (`Munch.fromDict(yaml.safe_load(...))`): the reference's YAML files load as they are.  Keys the path reads
    are validated, upstream keys are kept but unused, an unknown key is an error (a typo must not silently fall back)."""
    cfg = Cfg(yaml.safe_load(open(cfg_file)))
    cfg.update(overrides or {})
    unknown = [k for k in cfg if k not in SUPPORTED_KEYS and k not in UPSTREAM_KEYS]
    if unknown:
        raise KeyError(f"{cfg_file}: unknown config keys {unknown}")
    if 'texture_gen_method' not in cfg:
        raise KeyError(f"{cfg_file}: texture_gen_method is required")
    for k, v in DEFAULTS.items():
        cfg.setdefault(k, v)
    # camera_utils.py:116-245: 'blender' / 'exact_blender' always place 20 cameras, 'self_defined' knows 6 or 20 -- the per-view
    # arrays of the path are sized by view_num, so a mismatch is an error here instead of an index error later
    if cfg.camera_distribution in ('blender', 'exact_blender') and cfg.view_num != 20:
        raise ValueError(f"camera_distribution={cfg.camera_distribution!r} places 20 cameras: set view_num: 20 (got {cfg.view_num})")
    if cfg.optimize_from == 'None':                      # YAML `None` is the string 'None' (demo.py:213 treats both alike)
        cfg['optimize_from'] = None
    return cfg


def prepare(cfg_file, device, ckpt_path=None, allow_random_weights=False, overrides=None, batch_shapes=1):
    """demo.py:311-356 without the POCO network."""
    cfg = load_config(cfg_file, overrides)
    rank = int(os.environ.get('RANK', 0))
    stamp = datetime.datetime.now().strftime("%Y.%m.%d.%H.%M.%S") + (f'_rank{rank}' if int(os.environ.get('WORLD_SIZE', 1)) > 1 else '')
    logger = get_logger(os.path.join(cfg.output_path, f'{stamp}_log.log'))
    inpainter = None
    if cfg.texture_gen_method == 'DDNM_inpaint':
        logger.info('Loading inpainter...')
        from .ddnm_inpainting import Inpainter, DEFAULT_CKPT
        inpainter = Inpainter(device, ckpt_path=ckpt_path or DEFAULT_CKPT, allow_random_weights=allow_random_weights,
                              max_batch=cfg.view_num * max(1, int(batch_shapes)))
        logger.info('inpainter loaded')
    cams, base_dirs, eye_positions, up_dirs = create_cameras(num_views=cfg.view_num, distribution=cfg.camera_distribution,
                                                             distance=1.6, res=cfg.cam_res, device=device)
    camera_info = dict(cams=cams, base_dirs=base_dirs, eye_positions=eye_positions, up_dirs=up_dirs)
    return cfg, inpainter, camera_info, logger


_STANDIN = {}


def standin_geometry(xyz, atlas_res, device, logger):
    """No mesh next to the PLY and no POCO here: UV sphere (radius 0.5 = the normalised cloud's half extent) + analytic atlas.
    The same for every cloud, so a directory run builds it once per (atlas_res, device)."""
    logger.warning('no <pc>_untextured_mesh.obj supplied: using the stand-in UV sphere geometry (POCO/SPR/xatlas are upstream)')
    key = (int(atlas_res), str(device))
    if key not in _STANDIN:
        _STANDIN[key] = _standin_geometry(atlas_res, device)
    v, f, d = _STANDIN[key]
    return v.clone(), f.clone(), {k: t.clone() for k, t in d.items()}


def _standin_geometry(atlas_res, device):
    verts, faces, lut = synthetic.uv_sphere(50, 100)
    gb_pos, mask, fid = synthetic.latlong_atlas(atlas_res, 50, 100, lut=lut)
    # per-corner UVs of the lat-long parametrisation (one vt per face corner, like xatlas' unshared output)
    v = verts.astype(np.float64)
    lat = np.arccos(np.clip(v[:, 1] / 0.5, -1, 1)) / np.pi
    lon = (np.arctan2(v[:, 2], v[:, 0]) % (2 * np.pi)) / (2 * np.pi)
    corner_uv = np.stack([lon[faces], lat[faces]], -1)                          # [F,3,2]
    wrap = (corner_uv[..., 0].max(1) - corner_uv[..., 0].min(1)) > 0.5           # faces crossing the seam
    cu = corner_uv[..., 0]
    cu[wrap] = np.where(cu[wrap] < 0.5, cu[wrap] + 1.0, cu[wrap])
    span, g = atlas_res - 4, 2
    uvs = np.stack([(np.clip(cu, 0, 1) * span + g) / atlas_res, (corner_uv[..., 1] * span + g) / atlas_res], -1).reshape(-1, 2)
    tex_idx = np.arange(faces.shape[0] * 3, dtype=np.int64).reshape(-1, 3)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return T(verts), T(faces), dict(uvs=T(uvs.astype(np.float32)), mesh_tex_idx=T(tex_idx), gb_pos=T(gb_pos), mask=T(mask),
                                    per_atlas_pixel_face_id=T(fid))


def save_textured_mesh(vertices, uvs, faces, mesh_tex_idx, atlas_img, mask, output_root_path):
    """demo.py:264-307."""
    io_utils.savemeshtes2(vertices.cpu().numpy(), uvs.cpu().numpy(), faces.cpu().numpy(), mesh_tex_idx.cpu().numpy(),
                          os.path.join(output_root_path, 'models', 'model_normalized.obj'))
    # atlas [A,A,3] -> flipped vertically, uint8(clip(x*255)), PNG (device conversion + native encoder)
    io_utils.save_CHW_RGB_img(atlas_img.flip(0).permute(2, 0, 1), os.path.join(output_root_path, 'models', 'model_normalized.png'))
    # others/atlas_wo_background.png: the atlas with the chart mask as alpha (demo.py:296-303)
    rgba = torch.cat([atlas_img.float(), mask[0].to(atlas_img.device).float()], dim=-1)
    io_utils.save_CHW_RGBA_img(rgba.flip(0).permute(2, 0, 1), os.path.join(output_root_path, 'others', 'atlas_wo_background.png'))


def _load_shape(cfg, pc_file, name, device, logger):
    """demo.py:359-452: cloud, mesh and atlas of one shape, normalised as the reference does."""
    out = os.path.join(cfg.output_path, name)
    for d in ('geo', 'models', 'others'):
        os.makedirs(os.path.join(out, d), exist_ok=True)
    xyz, rgb = io_utils.read_ply_xyzrgb(pc_file)
    if len(xyz) > 30000:
        print(f"Point number > 30000! ({len(xyz)} points)({pc_file}) \n Please try uniformly subsampling the input point cloud first")
        raise NotImplementedError
    xyz = torch.tensor(np.asarray(xyz, np.float32)).to(device)
    rgb = torch.tensor(np.asarray(rgb)).float().to(device) / 255.0
    vmin, vmax = xyz.min(0)[0], xyz.max(0)[0]
    xyz -= (vmax + vmin) / 2.
    xyz /= (vmax - vmin).max()
    io_utils.save_colored_pc_ply(xyz.cpu().numpy(), rgb.cpu().numpy(), os.path.join(out, 'input_pc.ply'))
    geo_path = pc_file.replace('.ply', '_untextured_mesh.obj')
    xatlas_file = os.path.join(out, 'geo', f'xatlas_{cfg.xatlas_texture_res}.pth')
    if os.path.exists(geo_path):
        v, f, vt, ft = io_utils.load_obj_mesh(geo_path, with_uv=True)
        vertices = torch.from_numpy(v).to(device)
        faces = torch.from_numpy(f).to(device)
        vertices -= (vmax + vmin) / 2.
        vertices /= (vmax - vmin).max()
        if os.path.exists(xatlas_file):
            xatlas_dict = {k: (t.to(device) if torch.is_tensor(t) else t) for k, t in torch.load(xatlas_file).items()}
        elif vt is not None:
            # the mesh carries its UV parametrisation (`vt` + `f v/vt`): run the atlas producer (the rasterise + interpolate half of
            # xatlas_uvmap_w_face_id, extract_texture_map.py:48-64) and cache the dict in the reference's wire format (demo.py:445-448)
            from .extract_texture_map import uvmap_w_face_id
            uvs, tex_idx, gb_pos, amask, fid = uvmap_w_face_id(vertices, faces, torch.from_numpy(vt).to(device),
                                                               torch.from_numpy(ft).to(device), cfg.xatlas_texture_res)
            xatlas_dict = dict(uvs=uvs, mesh_tex_idx=tex_idx, gb_pos=gb_pos, mask=amask, per_atlas_pixel_face_id=fid)
            torch.save({k: t.cpu() for k, t in xatlas_dict.items()}, xatlas_file)
            logger.info(f'UV atlas rasterised from the mesh\'s own vt/f records -> {xatlas_file}')
        else:
            raise FileNotFoundError(f"{xatlas_file} not found and {geo_path} has no vt / f v/vt records: the chart parametrisation "
                                    "(xatlas.parametrize, CPU third-party) is upstream of this build; provide either")
        logger.info('Existing geometry + xatlas data loaded')
    else:
        vertices, faces, xatlas_dict = standin_geometry(xyz, cfg.xatlas_texture_res, device, logger)
    f_normals = torch.from_numpy(synthetic.face_normals(vertices.cpu().numpy(), faces.cpu().numpy())).to(device)
    return dict(out=out, coords=xyz, colors=rgb, vertices=vertices, faces=faces, f_normals=f_normals, xatlas=xatlas_dict)


def _pipeline_kwargs(cfg):
    kw = {k: cfg[k] for k in SUPPORTED_KEYS if k in cfg and k != 'output_path'}
    kw.pop('camera_distribution', None)
    return kw


def recon_one_textured_mesh(cfg, inpainter, camera_info, pc_file, name, device, logger):
    """demo.py:359-466 (texturing part)."""
    all_start = time.time()
    sh = _load_shape(cfg, pc_file, name, device, logger)
    logger.info('Generate texture by PointDreamer...')
    start = time.time()
    vertices, uvs, faces, mesh_tex_idx, atlas_img, mask = pipeline.colorize_one_mesh(
        sh['coords'], sh['colors'], sh['vertices'], sh['faces'], sh['f_normals'], sh['xatlas'], camera_info, inpainter=inpainter,
        save_img_path=os.path.join(sh['out'], 'others'), device=device, logger=None, **_pipeline_kwargs(cfg))
    torch.cuda.synchronize()
    logger.info(f'generate texture time: {time.time() - start} s')
    save_textured_mesh(vertices, uvs, faces, mesh_tex_idx, atlas_img, mask, sh['out'])
    logger.info(f'total time: {time.time() - all_start} s')
    return sh['out']


def recon_textured_meshes_batched(cfg, inpainter, camera_info, pc_files, names, device, logger):
    """A directory run, several clouds at a time: the same stages and files as recon_one_textured_mesh per shape, with the views
    of all the shapes of the group in one inpainter batch (pipeline.colorize_meshes_batched)."""
    all_start = time.time()
    shapes = [_load_shape(cfg, pc, name, device, logger) for pc, name in zip(pc_files, names)]
    logger.info(f'Generate texture by PointDreamer ({len(shapes)} shapes in one batch)...')
    start = time.time()
    results = pipeline.colorize_meshes_batched(shapes, camera_info, inpainter=inpainter, return_full=True,
                                               save_img_paths=[os.path.join(sh['out'], 'others') for sh in shapes],
                                               **_pipeline_kwargs(cfg))
    torch.cuda.synchronize()
    logger.info(f'generate texture time: {time.time() - start} s ({(time.time() - start) / len(shapes)} s per shape)')
    for sh, (vertices, uvs, faces, mesh_tex_idx, atlas_img, mask) in zip(shapes, results):
        save_textured_mesh(vertices, uvs, faces, mesh_tex_idx, atlas_img, mask, sh['out'])
    logger.info(f'total time: {time.time() - all_start} s')
    return [sh['out'] for sh in shapes]


def recon_one_textured_mesh_view_parallel(cfg, inpainter, camera_info, pc_file, name, device, logger, rank, world, shape_key):
    """`--parallel views`: the views of ONE shape are split over the ranks (dist.colorize_one_mesh_view_parallel);
    every rank writes the per-view PNGs of its own views, rank 0 the mesh / atlas files."""
    from . import dist as pdist
    all_start = time.time()
    sh = _load_shape(cfg, pc_file, name, device, logger) if rank == 0 else None
    if world > 1:
        torch.distributed.barrier()                      # rank 0 created the output tree (input_pc.ply, geo cache) first
        if rank != 0:
            sh = _load_shape(cfg, pc_file, name, device, logger)
    logger.info(f'Generate texture by PointDreamer (views of the shape split over {world} ranks)...')
    start = time.time()
    vertices, uvs, faces, mesh_tex_idx, atlas_img, mask = pdist.colorize_one_mesh_view_parallel(
        sh['coords'], sh['colors'], sh['vertices'], sh['faces'], sh['f_normals'], sh['xatlas'], camera_info, rank=rank, world=world,
        inpainter=inpainter, save_img_path=os.path.join(sh['out'], 'others'), shape_key=shape_key, return_full=True,
        **_pipeline_kwargs(cfg))
    torch.cuda.synchronize()
    logger.info(f'generate texture time: {time.time() - start} s')
    if rank == 0:
        save_textured_mesh(vertices, uvs, faces, mesh_tex_idx, atlas_img, mask, sh['out'])
    logger.info(f'total time: {time.time() - all_start} s')
    return sh['out']


def files_of_rank(pc_files, rank, world):
    """The clouds rank `rank` of `world` textures: a contiguous block of the sorted list (dist.shard_range)."""
    from .dist import shard_range
    return [pc_files[i] for i in shard_range(len(pc_files), rank, world)]


def main(argv=None):
    p = argparse.ArgumentParser("PointDreamer (MI355X texturing path)")
    p.add_argument("--config", type=str, default='configs/default.yaml', help="path to config file")
    p.add_argument("--pc_file", type=str, default='dataset/demo_data/clock.ply', help="path to input point cloud file or directory")
    p.add_argument("--ckpt", type=str, default=None, help="256x256_diffusion_uncond.pt (reference key names)")
    p.add_argument("--allow_random_weights", action='store_true', help="run DDNM with random-init weights if the checkpoint is absent")
    p.add_argument("--set", nargs='*', default=[], help="YAML overrides key=value (e.g. complete_unseen_by=unproject optimize_from=None)")
    p.add_argument("--batch_shapes", type=int, default=4, help="directory runs: clouds textured together, their views in one "
                   "inpainter batch (1 = one at a time, as the reference -- the only route that re-uses existing {k}_inpainted.png "
                   "files of a resumed output directory, demo.py:138-147; 4 is ~16 %% more shapes/hour on one MI355X)")
    p.add_argument("--parallel", choices=['shapes', 'views'], default='shapes', help="under torch.distributed.run: 'shapes' = every "
                   "rank textures its own block of the clouds (no collective); 'views' = the views of each shape are split over the "
                   "ranks and assembled with one all_gather (lowest latency per shape; needs world size <= view_num)")
    args = p.parse_args(argv)
    # Host threads: torch sizes its intra-op pool by the machine's logical CPUs (256 on the pool's boxes) while the container's cgroup
    # grants 16 -- a 128-thread OpenMP region around a 90 000-element `.float()` next to the PNG-encoder threads exhausts the CFS quota
    # and the whole process is throttled for the rest of the period (round 5: 99 ms per `_load_shape` in a directory run against 2.5 ms
    # alone).  The CLI's CPU tensors are small: a few threads, inside the quota.
    torch.set_num_threads(max(1, min(4, io_utils.cpus_per_rank() // 4)))
    # one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m pointdreamer_amd.demo ...`): every rank textures its
    # own contiguous block of the directory's clouds -- independent shapes, no collective on the data path
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0))) if world > 1 else torch.device('cuda')
    if world > 1:
        torch.cuda.set_device(device)
    overrides = {k: yaml.safe_load(v) for k, v in (kv.split('=', 1) for kv in args.set)}
    pc_files = [args.pc_file] if args.pc_file.endswith('.ply') else \
        [os.path.join(args.pc_file, i) for i in sorted(os.listdir(args.pc_file)) if i.endswith('.ply')]
    by_views = args.parallel == 'views' and world > 1
    if by_views:
        import torch.distributed as tdist
        if not tdist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            tdist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    else:
        pc_files = files_of_rank(pc_files, rank, world)
    group = 1 if by_views else max(1, min(args.batch_shapes, max(1, len(pc_files))))
    cfg, inpainter, camera_info, logger = prepare(args.config, device, args.ckpt, args.allow_random_weights, overrides, batch_shapes=group)
    outs = []
    # PNG / OBJ encoding of one shape runs on host threads under the GPU work of the next (io_utils.set_async); every file is on
    # disk when main() returns
    io_utils.set_async(True, workers=max(2, min(8, io_utils.cpus_per_rank() - 4)))
    try:
        for g0 in range(0, len(pc_files), group):
            chunk = pc_files[g0:g0 + group]
            names = [os.path.basename(f).split('.ply')[0] + '_' + os.path.basename(args.config).split('.')[0] for f in chunk]
            for pc_file, name in zip(chunk, names):
                os.makedirs(os.path.join(cfg.output_path, name), exist_ok=True)
                shutil.copy(args.config, os.path.join(cfg.output_path, name, 'config.yaml'))
                logger.info(f'Start Recon {pc_file}...')
            if by_views:
                outs.append(recon_one_textured_mesh_view_parallel(cfg, inpainter, camera_info, chunk[0], names[0], device, logger,
                                                                  rank, world, shape_key=g0))
            elif len(chunk) == 1:
                outs.append(recon_one_textured_mesh(cfg, inpainter, camera_info, chunk[0], names[0], device, logger))
            else:
                outs += recon_textured_meshes_batched(cfg, inpainter, camera_info, chunk, names, device, logger)
    finally:
        io_utils.set_async(False)          # flushes
    return outs


if __name__ == '__main__':
    main()
