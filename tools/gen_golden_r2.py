"""Round-2 golden vectors, again by RUNNING THE REFERENCE (imported from /root/reference, CPU tensors).
Build container only:  python -m tools.gen_golden_r2 [p1 o1 triptych clock]

  p1_crop.npz        ours_utils.get_rendered_hard_mask_and_face_idx_batch (:93-150): the crop / rescale arithmetic of row P1
                     (uv_centers, uv_scales, vertice_uvs, point_uvs, point_depths and the `pos` handed to the rasteriser) computed
                     by the reference itself; only cam.transform (kaolin) and nvdiffrast.rasterize are stand-ins (the oracle's).
  o1_writers.npz     utils_3d.savemeshtes2 (:27-64) OBJ / MTL bytes, utils_2d.save_CHW_RGB(A)_img (:351-381) and
                     demo.save_textured_mesh (:264-307) PNG pixels (decoded), written by the reference's own functions.
  triptych.npz       unproject.get_shrinked_per_view_per_pixel_visibility_torch(save_path=...) (:429-475) with the reference's
                     cat_images / save_CHW_RGB_img: the decoded `shrink_per_view_edge/{v}.png` files.
  clock_nearest.npz  BASELINE configs[0]: dataset/demo_data/clock.ply through the reference's demo.colorize_one_mesh with
                     configs/nearest.yaml values (texture_gen_method nearest, hidden-point removal on, NBF [21],
                     complete_unseen_by neighbor) on CPU.  Stand-ins for the absent third-party packages: kaolin camera
                     (oracle/camera.py), nvdiffrast.rasterize (oracle/project.py), open3d hidden_point_removal (oracle: scipy
                     qhull), torchvision Resize, kaolin sided_distance / uniform_laplacian, trimesh helpers (tools/ref_harness.py).
                     The geometry is the build's stand-in UV sphere (POCO / xatlas are upstream of the path).  `optimize_from` is off
                     here (exact stages only); the optimisation loop is pinned on its own by optimize_*.npz below.
  optimize_*.npz     SURVEY 8f-1: pointdreamer.ours_utils.optimize_color (:1583-1785) -- the reference's own Adam(5e-2) / StepLR(15, 0.5) loop,
                     f64 bilinear lookup, L1 loss masked by foreground and shrunk visibility -- run here on the CPU with V = 3 views at its
                     hard-wired 1024^2 render size, atlas 64^2 and 128^2, 3 and 100 iterations, with and without the shrunk visibility.
                     Stand-ins: nvdiffrast.rasterize / interpolate (the oracle's rasteriser + barycentric interpolation),
                     kaolin.render.mesh.texture_mapping (grid_sample, align_corners=False, border: its published behaviour), kaolin's
                     camera-matrix helpers and prepare_vertices (their results are unused on the nvdiffrast branch), torchvision Resize,
                     and `device='cuda'` in one torch.ones call of the unused attribute list (dropped).
Inputs that come from the build's generators are stored as inputs; every `ref_*` array was computed by reference code."""
import io
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_harness as rh                      # noqa: E402
from tools.gen_golden import TorchCam, scene             # noqa: E402
from oracle import camera as ocam, project as oproj       # noqa: E402
from pointdreamer_amd import synthetic                     # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
t = torch.from_numpy


def install_raster_stub():
    """nvdiffrast.torch.rasterize(glctx, pos, tri, resolution) -> (rast,), rast[..., 2] = z/w, rast[..., 3] = face id + 1:
    the oracle's rasteriser in nvdiffrast's output layout (the fill rule itself is third-party, parity unpinned)."""
    nv = sys.modules['nvdiffrast'].torch
    seen = {}

    def rasterize(glctx, pos, tri, resolution, grad_db=False, **kw):
        p = pos.detach().cpu().numpy().astype(np.float32)
        hard, fid, depth = oproj.rasterize(p, tri.detach().cpu().numpy().astype(np.int64), int(resolution[0]))
        rast = np.zeros(fid.shape + (4,), np.float32)
        rast[..., 2] = depth
        rast[..., 3] = (fid + 1).astype(np.float32)
        seen['pos'] = p
        return (torch.from_numpy(rast), None)
    nv.rasterize = rasterize
    sys.modules['nvdiffrast'].torch = nv
    return seen


def gen_p1():
    ou, up, u2 = rh.import_reference()
    seen = install_raster_stub()
    ou.nvdiffrast = sys.modules['nvdiffrast']
    out = {}
    for tag, (npts, seed, V, R, rescale, pad) in dict(a=(2000, 1, 3, 128, True, 0.05), b=(700, 7, 4, 96, True, 0.1),
                                                       c=(500, 3, 2, 64, False, 0.05)).items():
        verts, faces, _ = synthetic.uv_sphere(12, 24)
        verts = (verts * np.array([1.0, 0.7, 1.3], np.float32)).astype(np.float32)      # not a sphere: anisotropic crop boxes
        xyz, rgb = synthetic.sphere_points(npts, seed=seed)
        xyz = (xyz * np.array([1.0, 0.7, 1.3], np.float32)).astype(np.float32)
        cams, base_dirs, eyes, ups = ocam.create_cameras(V, 1.6, R)
        tc = [TorchCam(c) for c in cams]
        r = ou.get_rendered_hard_mask_and_face_idx_batch(tc, t(verts), t(faces), t(xyz), None, rescale=rescale, padding=pad)
        hard, fidx, depth, vuv, uvc, uvs, padding, puv, pdep = r
        out.update({f'{tag}_cam_params': np.stack([c.params for c in cams]), f'{tag}_vertices': verts, f'{tag}_faces': faces,
                    f'{tag}_points': xyz, f'{tag}_cam_res': R, f'{tag}_rescale': rescale, f'{tag}_padding': pad,
                    f'{tag}_ref_hard': hard.numpy(), f'{tag}_ref_face_idx': fidx.numpy(), f'{tag}_ref_depth': depth.numpy(),
                    f'{tag}_ref_vertice_uvs': vuv.numpy(), f'{tag}_ref_point_uvs': puv.numpy(), f'{tag}_ref_point_depths': pdep.numpy(),
                    f'{tag}_ref_pos': seen['pos'],
                    f'{tag}_ref_uv_centers': uvc.numpy() if torch.is_tensor(uvc) else np.float32(uvc),
                    f'{tag}_ref_uv_scales': uvs.numpy() if torch.is_tensor(uvs) else np.float32(uvs),
                    f'{tag}_ref_padding': np.float32(padding)})
    np.savez_compressed(os.path.join(OUT, 'p1_crop.npz'), **out)


def _decode(path):
    import PIL.Image
    im = PIL.Image.open(path)
    return im.mode, np.array(im)


def gen_o1():
    rh.install()
    cwd = os.getcwd()
    os.chdir(rh.REF)
    try:
        from models.get3d.get3d_utils.utils_3d import savemeshtes2
        import utils.utils_2d as u2
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(11)
    verts, faces, _ = synthetic.uv_sphere(6, 9)
    verts = (verts.astype(np.float64) * 3.7 + rng.normal(0, 1e-4, verts.shape)).astype(np.float32)     # many digits, negative zeros, ...
    verts[0] = [-0.0, 1e-7, -123456.789]
    verts[1] = [0.5, -2.5e-7, 1e6]
    uvs, fuv = synthetic.uv_sphere_uvs(6, 9, 64, gutter=2)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        savemeshtes2(verts.copy(), uvs.copy(), faces.copy(), fuv.copy(), os.path.join(d, 'model_normalized.obj'))
        out['ref_obj'] = np.frombuffer(open(os.path.join(d, 'model_normalized.obj'), 'rb').read(), np.uint8)
        out['ref_mtl'] = np.frombuffer(open(os.path.join(d, 'model_normalized.mtl'), 'rb').read(), np.uint8)
        img3 = rng.uniform(-0.2, 1.2, (3, 37, 53)).astype(np.float32)          # out-of-range values exercise the clip
        img3[:, 0, :8] = np.array([0.0, 1.0, 0.5, 0.49999, 254.5 / 255, 254.999 / 255, 1e-8, 0.999999], np.float32)
        img4 = rng.uniform(-0.2, 1.2, (4, 29, 31)).astype(np.float32)
        u2.save_CHW_RGB_img(img3.copy(), os.path.join(d, 'rgb.png'))
        u2.save_CHW_RGBA_img(img4.copy(), os.path.join(d, 'rgba.png'))
        m3, p3 = _decode(os.path.join(d, 'rgb.png'))
        m4, p4 = _decode(os.path.join(d, 'rgba.png'))
        assert m3 == 'RGB' and m4 == 'RGBA'
        out.update(vertices=verts, uvs=uvs, faces=faces, face_uv_idx=fuv, img_rgb=img3, img_rgba=img4, ref_rgb_pixels=p3,
                   ref_rgba_pixels=p4)
        back = u2.load_CHW_RGB_img(os.path.join(d, 'rgba.png'))
        out['ref_loaded_from_rgba'] = back.numpy()
        # demo.save_textured_mesh: atlas [A,A,3] float + chart mask -> models/model_normalized.png (flipped), others/atlas_wo_background.png
        demo = import_reference_demo()
        A = 48
        atlas = rng.uniform(-0.1, 1.1, (A, A, 3)).astype(np.float32)
        mask = (rng.uniform(0, 1, (1, A, A, 1)) > 0.4)
        os.makedirs(os.path.join(d, 'models')); os.makedirs(os.path.join(d, 'others'))
        demo.save_textured_mesh(t(verts), t(uvs), t(faces), t(fuv), t(atlas.copy()), t(mask), d)
        out.update(atlas=atlas, atlas_mask=mask, ref_atlas_png=_decode(os.path.join(d, 'models', 'model_normalized.png'))[1],
                   ref_atlas_rgba_png=_decode(os.path.join(d, 'others', 'atlas_wo_background.png'))[1],
                   ref_obj2=np.frombuffer(open(os.path.join(d, 'models', 'model_normalized.obj'), 'rb').read(), np.uint8))
    np.savez_compressed(os.path.join(OUT, 'o1_writers.npz'), **out)


def gen_triptych():
    rh.install()
    cwd = os.getcwd()
    os.chdir(rh.REF)
    try:
        import pointdreamer.unproject as up
        import utils.utils_2d as u2
    finally:
        os.chdir(cwd)
    import importlib
    importlib.reload(up)                                  # undo import_reference()'s no-op writers if they were installed
    rng = np.random.default_rng(5)
    A, V = 64, 3
    yy, xx = np.mgrid[0:A, 0:A]
    mask = ((yy - 30) ** 2 + (xx - 33) ** 2 < 27 ** 2) | (yy > 56)
    vis = np.stack([(mask & (xx < 20 + 9 * v + 4 * np.sin(yy / 5.0))) | (mask & (rng.uniform(0, 1, (A, A)) > 0.97)) for v in range(V)], -1)
    with tempfile.TemporaryDirectory() as d:
        pk = up.get_shrinked_per_view_per_pixel_visibility_torch(t(mask), t(vis), kernel_sizes=[7, 3], save_path=os.path.join(d, 'sp'))
        pngs = np.stack([_decode(os.path.join(d, 'sp', f'{v}.png'))[1] for v in range(V)])
    np.savez_compressed(os.path.join(OUT, 'triptych.npz'), mask=mask, vis=vis, kernels=np.array([7, 3]), ref_shrinked=pk.numpy(),
                        ref_pngs=pngs)


_DEMO = None


def import_reference_demo():
    """The reference's demo.py (its colorize_one_mesh / save_textured_mesh) with POCO / SPR -- geometry, upstream of the path --
    stubbed whole."""
    global _DEMO
    if _DEMO is not None:
        return _DEMO
    rh.install()
    for n in ['baselines', 'baselines.spr']:
        sys.modules.setdefault(n, rh._Stub(n))
    sys.modules['baselines'].spr = sys.modules['baselines.spr']
    cwd = os.getcwd()
    os.chdir(rh.REF)
    try:
        import models.POCO  # noqa: F401  (namespace package)
        sys.modules['models.POCO.generate_1'] = rh._Stub('models.POCO.generate_1')
        import demo
    finally:
        os.chdir(cwd)
    _DEMO = demo
    return demo


def install_o3d_stub():
    o3d = sys.modules['open3d']

    class PointCloud:
        def __init__(self, points=None):
            self.points = np.asarray(points, np.float64)

        def hidden_point_removal(self, camera, radius):
            vis = oproj.point_validation_by_hpr(self.points, [np.asarray(camera, np.float64)], radius)[0]
            return None, np.nonzero(vis)[0]
    o3d.geometry.PointCloud = PointCloud
    o3d.utility.Vector3dVector = lambda a: np.asarray(a, np.float64)


def gen_clock():
    import yaml
    torch.set_num_threads(1)                                # the reference's duplicate-index writes: last write wins
    demo = import_reference_demo()
    install_raster_stub()
    install_o3d_stub()
    import pointdreamer.ours_utils as rou
    import pointdreamer.unproject as rup
    rup.cat_images = lambda a, b, *k, **kw: a               # the debug triptychs are pinned by triptych.npz
    rup.save_CHW_RGB_img = lambda *a, **k: None
    cfg = yaml.safe_load(open(os.path.join(rh.REF, 'configs', 'nearest.yaml')))
    from pointdreamer_amd import io_utils
    import pointdreamer_amd.demo as pdemo
    xyz, rgb = io_utils.read_ply_xyzrgb(os.path.join(rh.REF, 'dataset', 'demo_data', 'clock.ply'))
    # demo.py:370-378 (normalisation) in torch, as the reference does it
    X = t(xyz.copy())
    C = t(rgb.copy()).float() / 255.0
    vmin, vmax = X.min(0)[0], X.max(0)[0]
    X -= (vmax + vmin) / 2.
    X /= (vmax - vmin).max()
    A = int(cfg['xatlas_texture_res'])
    verts, faces, xd = pdemo._standin_geometry(A, 'cpu')
    cams, base_dirs, eyes, ups = ocam.create_cameras(cfg['view_num'], 1.6, cfg['cam_res'])
    tc = [TorchCam(c) for c in cams]
    fn = t(synthetic.face_normals(verts.numpy(), faces.numpy()))
    camera_info = dict(cams=tc, base_dirs=t(np.asarray(base_dirs, np.float32)), eye_positions=eyes, up_dirs=ups, cam_RTs=None, cam_K=None)
    cap = {}
    real_unproject = demo.unproject

    def spy_unproject(*a, **k):
        r = real_unproject(*a, **k)
        cap['inpainted'] = a[0].detach().clone().numpy()
        cap['atlas_unprojected'] = r[0].detach().clone().numpy()
        cap['shrinked'] = r[1].detach().clone().numpy()
        cap['view_ids'] = r[2].detach().clone().numpy()
        cap['painted'] = r[5].detach().clone().numpy()
        return r
    demo.unproject = spy_unproject
    real_sparse = demo.get_sparse_images

    def spy_sparse(*a, **k):
        r = real_sparse(*a, **k)
        cap['point_validation'] = a[2].detach().clone().numpy()
        cap['sparse'], cap['mask0'], cap['mask2'], cap['scale_factors'] = (x.detach().clone().numpy() for x in r)
        return r
    demo.get_sparse_images = spy_sparse
    kw = dict(cfg)
    kw['optimize_from'] = None
    with tempfile.TemporaryDirectory() as d:
        out = demo.colorize_one_mesh(X.clone(), C.clone(), verts.clone(), faces.clone(), fn, {k: v.clone() for k, v in xd.items()},
                                     camera_info, device='cpu', save_img_path=d, inpainter=None, glctx=None, logger=None, **kw)
    atlas = out[4].detach().numpy()
    np.savez_compressed(
        os.path.join(OUT, 'clock_nearest.npz'),
        ref_point_validation=np.packbits(cap['point_validation'], axis=1), n_points=np.int64(X.shape[0]),
        ref_sparse_u8=(cap['sparse'] * 255.0).round().astype(np.uint8),       # colours are uint8 / 255: exact in 8 bits
        ref_mask0=np.packbits(cap['mask0'][:, 0] > 0, axis=2), ref_mask2=np.packbits(cap['mask2'][:, 0] > 0, axis=2),
        ref_scale_factors=cap['scale_factors'], ref_inpainted_u8=(cap['inpainted'] * 255.0).round().astype(np.uint8),
        ref_view_ids=cap['view_ids'].astype(np.int8), ref_painted=np.packbits(cap['painted'], axis=1),
        ref_shrinked=np.packbits(cap['shrinked'], axis=2),
        ref_atlas_unprojected_u8=(cap['atlas_unprojected'] * 255.0).round().astype(np.uint8),
        ref_atlas_u8=np.clip(atlas * 255.0, 0, 255).astype(np.uint8), ref_atlas_f16=atlas.astype(np.float16))
    import shutil
    shutil.copy(os.path.join(rh.REF, 'dataset', 'demo_data', 'clock.ply'), os.path.join(OUT, 'clock.ply'))
    print('clock: visible fraction', cap['point_validation'].mean(), 'painted', cap['painted'].mean(), 'atlas mean', atlas.mean())


def gen_optimize():
    """optimize_{A}_{its}_{shr}.npz: inputs + the atlas the REFERENCE's optimize_color returns (ours_utils.py:1583-1785)."""
    import types
    import torch.nn.functional as F
    ou, up, u2 = rh.import_reference()
    seen = install_raster_stub()
    nv = sys.modules['nvdiffrast'].torch

    def interpolate(attr, rast, tri, **kw):              # nvdiffrast.torch.interpolate(uvs, rast, face_uvs_idx) -> (values,)
        fid = rast[..., 3].long().numpy() - 1
        bary = oproj.raster_barycentrics(seen['pos'], seen['faces'], fid, fid.shape[1])
        return (torch.from_numpy(oproj.interpolate(attr.detach().cpu().numpy().astype(np.float32), tri.detach().cpu().numpy().astype(np.int64), fid, bary)),)
    _rast = nv.rasterize

    def rasterize(glctx, pos, tri, resolution, grad_db=False, **kw):
        seen['faces'] = tri.detach().cpu().numpy().astype(np.int64)
        return _rast(glctx, pos, tri, resolution, grad_db=grad_db, **kw)
    nv.rasterize, nv.interpolate = rasterize, interpolate
    ou.dr = nv; ou.nvdiffrast = sys.modules['nvdiffrast']
    kal = sys.modules['kaolin']

    def texture_mapping(texture_coords, atlas, mode='bilinear'):   # kaolin.render.mesh.texture_mapping: uv in [0,1], v up -> [B,H,W,C]
        g = texture_coords * 2.0 - 1.0
        g = torch.stack([g[..., 0], -g[..., 1]], -1)
        return F.grid_sample(atlas, g, mode=mode, align_corners=False, padding_mode='border').permute(0, 2, 3, 1)
    kal.render.mesh.texture_mapping = texture_mapping
    kal.render.mesh.prepare_vertices = lambda *a, **k: (None, None, None)
    cam_mod = sys.modules.get('kaolin.render.camera') or kal.render.camera
    cam_mod.generate_transformation_matrix = lambda e, l, u_: torch.zeros((len(e), 4, 3))
    cam_mod.generate_perspective_projection = lambda fovy, ratio=1.0: torch.zeros((3, 1))
    cam_mod.perspective_camera = lambda *a, **k: None
    sys.modules['kaolin.render.camera'] = cam_mod
    # `torch.ones(..., device='cuda')` (ours_utils.py:1671, an attribute list the nvdiffrast branch never reads): drop the device
    proxy = types.ModuleType('torch_cpu_proxy')
    proxy.__dict__.update(torch.__dict__)
    proxy.ones = lambda *a, device=None, **k: torch.ones(*a, **k)
    ou.torch = proxy
    torch.manual_seed(0)
    V, R, r = 3, 128, 64
    verts, faces, lut = synthetic.uv_sphere(10, 20)
    cams, base_dirs, eyes, ups = ocam.create_cameras(V, 1.6, R)
    pr = oproj.project_batch(cams, verts, verts[:8], True, 0.05)
    uvs_np, tex_idx = synthetic.uv_sphere_uvs(10, 20, A=64, gutter=2)      # (what xatlas would hand over as `uvs`, `mesh_tex_idx`)
    sf = np.array([1.0, 0.9, 1.1], np.float32)
    tcams = [TorchCam(c) for c in cams]
    g = np.random.default_rng(5)
    for A, its, use_shr in ((64, 3, True), (64, 100, True), (64, 100, False), (128, 3, False), (128, 100, True)):
        atlas0 = g.random((3, A, A), dtype=np.float32)
        inp = g.random((V, 3, r, r), dtype=np.float32)
        shr = (g.random((V, A, A)) > 0.3) if use_shr else None
        atlas, images = ou.optimize_color(t(atlas0.copy()), t(inp.copy()), t(verts), t(faces), t(uvs_np), t(tex_idx), tcams, t(eyes), None, t(ups),
                                          t(pr['uv_centers']), t(pr['uv_scales']), 0.05, t(sf), None,
                                          shrinked_per_view_per_pixel_visibility=None if shr is None else t(shr), lr=5e-2, iterations=its)
        images = images.detach()
        np.savez_compressed(os.path.join(OUT, f'optimize_{A}_{its}_{int(use_shr)}.npz'), atlas0=atlas0, inpainted=inp, verts=verts, faces=faces, uvs=uvs_np,
                            mesh_tex_idx=tex_idx, cam_params=np.stack([c.params for c in cams]), cam_res=np.int64(R), uv_centers=pr['uv_centers'],
                            uv_scales=pr['uv_scales'], padding=np.float32(0.05), scale_factors=sf, shrinked=np.zeros((0,), bool) if shr is None else shr,
                            iterations=np.int64(its), ref_atlas=atlas.detach().numpy().astype(np.float32),
                            ref_images_mean=images.mean(dim=(2, 3)).numpy(), ref_images_small=images[:, :, ::16, ::16].numpy().astype(np.float32))
        print('optimize', A, its, use_shr, float((atlas.detach().numpy() - atlas0).__abs__().max()))


if __name__ == '__main__':
    assert rh.available()
    which = sys.argv[1:] or ['p1', 'o1', 'triptych', 'clock', 'optimize']
    os.makedirs(OUT, exist_ok=True)
    for w in which:
        dict(p1=gen_p1, o1=gen_o1, triptych=gen_triptych, clock=gen_clock, optimize=gen_optimize)[w]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
