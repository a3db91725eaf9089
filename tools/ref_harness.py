"""Import harness for the reference (build-container only; /root/reference does not travel).

Injects stub modules for the un-installed third-party packages the reference imports at module
scope (kaolin, nvdiffrast, torchvision, open3d, cv2, trimesh, ...), then imports the reference's
own modules unmodified from /root/reference.  Used ONLY by tools/gen_golden*.py to emit golden
vectors and by tests that are skipped when /root/reference is absent.  No reference source is
copied; the stubs carry no reference logic except two stand-ins for third-party calls:
  * torchvision.transforms.Resize -> F.interpolate(bilinear, align_corners=False, antialias=False)
    (what torchvision 0.15/0.16 does for tensors), bool tensors resized via float then `!= 0`;
  * kaolin.metrics.pointcloud.sided_distance -> exact brute force (first minimum).
  * kaolin.ops.mesh.uniform_laplacian, trimesh.grouping.unique_rows, trimesh.geometry.faces_to_edges -> restatements of
    their published behaviour (used by paint_invisible_areas_by_neighbors / subdivide_with_uv).
"""
import os
import sys
import types

REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'pointdreamer'))


class _Stub(types.ModuleType):
    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []

    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        child = _Stub(self.__name__ + '.' + k)
        setattr(self, k, child)
        return child

    def __call__(self, *a, **k):
        return None


_STUBS = ['kaolin', 'nvdiffrast', 'nvdiffrast.torch', 'torchvision', 'torchvision.transforms',
          'torchvision.transforms.transforms', 'torchvision.transforms.functional', 'torchvision.utils',
          'torchvision.datasets', 'torchvision.datasets.utils', 'open3d', 'cv2', 'trimesh',
          'trimesh.grouping', 'trimesh.geometry', 'imageio', 'pytz', 'xatlas', 'kiui', 'seaborn', 'plyfile',
          'mcubes', 'munch', 'matplotlib', 'matplotlib.pyplot', 'pymeshlab', 'lpips', 'skimage']


def install():
    import torch
    import torch.nn.functional as F
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in _STUBS:
        try:
            __import__(name)
            continue
        except Exception:
            pass
        if name in sys.modules and not isinstance(sys.modules[name], _Stub):
            continue
        m = _Stub(name)
        sys.modules[name] = m
        if '.' in name:
            parent, child = name.rsplit('.', 1)
            setattr(sys.modules[parent], child, m)

    class Resize:
        def __init__(self, size, *a, **k):
            self.size = size

        def __call__(self, img):
            size = tuple(int(s) for s in self.size)
            was_bool = img.dtype == torch.bool
            x = img.float()
            squeeze = x.dim() == 3
            if squeeze:
                x = x.unsqueeze(0)
            y = F.interpolate(x, size=size, mode='bilinear', align_corners=False, antialias=False)
            if squeeze:
                y = y.squeeze(0)
            return (y != 0) if was_bool else y

    class Pad:
        def __init__(self, padding, fill=0):
            self.p = padding
            self.fill = fill

        def __call__(self, img):
            px, py = int(self.p[0]), int(self.p[1])
            return F.pad(img, (px, px, py, py), value=self.fill)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    tt = sys.modules['torchvision.transforms.transforms']
    if isinstance(tt, _Stub):
        tt.Resize, tt.Pad, tt.Compose = Resize, Pad, Compose
        sys.modules['torchvision.transforms'].transforms = tt
        sys.modules['torchvision.transforms'].Resize = Resize

    def sided_distance(p1, p2):
        d = torch.cdist(p1.double(), p2.double()) ** 2
        v, i = d.min(-1)
        return v, i
    kal = sys.modules['kaolin']
    if isinstance(kal, _Stub):
        kal.metrics.pointcloud.sided_distance = sided_distance

        def uniform_laplacian(num_vertices, faces):
            """kaolin.ops.mesh.uniform_laplacian (kaolin 0.15 docs): dense [V,V], row i = 1/deg(i) on the (unique) neighbours of
            vertex i, -1 on the diagonal, rows of isolated vertices 0."""
            f = faces.reshape(-1, 3).long()
            e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
            e = torch.cat([e, e.flip(1)], 0).unique(dim=0)
            adj = torch.zeros((num_vertices, num_vertices), dtype=torch.float32, device=faces.device)
            adj[e[:, 0], e[:, 1]] = 1.0
            adj[torch.arange(num_vertices), torch.arange(num_vertices)] = 0.0
            L = adj / adj.sum(1, keepdim=True)
            L[torch.arange(num_vertices), torch.arange(num_vertices)] = -1.0
            L[torch.isnan(L)] = 0.0
            return L
        kal.ops.mesh.uniform_laplacian = uniform_laplacian
    tg = sys.modules.get('trimesh.grouping')
    if isinstance(tg, _Stub):
        import numpy as np

        def unique_rows(data, digits=None, keep_order=False):
            """trimesh.grouping.unique_rows for small non-negative integer rows (trimesh 4.x: rows are bit-packed into one int64,
            column j shifted by j*floor(64/ncols), then np.unique(return_index, return_inverse))."""
            d = np.asanyarray(data).astype(np.int64)
            if len(d) == 0:                                    # trimesh: hashable_rows([]) -> [], np.unique([]) -> empty index arrays
                return np.zeros((0,), np.int64), np.zeros((0,), np.int64)
            prec = 64 // d.shape[1]
            assert np.abs(d).max() < 2 ** (prec - 1)
            h = np.zeros(len(d), np.int64)
            for off, col in enumerate(d.T):
                h ^= col << (off * prec)
            _, unique, inverse = np.unique(h, return_index=True, return_inverse=True)
            return unique, inverse

        def faces_to_edges(faces, return_index=False):
            """trimesh.geometry.faces_to_edges: (n,3) faces -> (3n,2) edges (0-1, 1-2, 2-0 per face)."""
            f = np.asanyarray(faces)
            return f[:, [0, 1, 1, 2, 2, 0]].reshape((-1, 2))
        tg.unique_rows = unique_rows
        sys.modules['trimesh.geometry'].faces_to_edges = faces_to_edges


def import_reference():
    """Returns (ours_utils, unproject_mod, utils_2d)."""
    install()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import pointdreamer.ours_utils as ou
        import pointdreamer.unproject as up
        import utils.utils_2d as u2
    finally:
        os.chdir(cwd)
    # debug image writers inside unproject() are not part of the arithmetic
    up.cat_images = lambda a, b, *k, **kw: a
    up.save_CHW_RGB_img = lambda *a, **k: None
    return ou, up, u2


def import_reference_unet():
    install()
    from models.DDNM.guided_diffusion import unet, script_util, nn as gnn
    return unet, script_util, gnn
