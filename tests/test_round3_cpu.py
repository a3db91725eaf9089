"""Round-3 CPU tests: camera distributions against the imported reference's create_cameras (tools/gen_golden_r3.py)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("tag,dist", [("fib8", "fibonacci_sphere"), ("fib5", "fibonacci_sphere"), ("self6", "self_defined"),
                                       ("self20", "self_defined"), ("blender", "blender"), ("exact_blender", "exact_blender")])
def test_c0_camera_distributions_vs_reference(tag, dist):
    """Row C0, utils/camera_utils.py:116-245: eye positions, base directions, up vectors and the vertical fov of every
    `camera_distribution` the reference accepts ('blender' / 'exact_blender' always give the 20 dodecahedron views)."""
    import pointdreamer_amd.camera_utils as cu
    g = load_golden('camera_distributions.npz')
    nv, d = int(g[tag + '_args'][0]), float(g[tag + '_args'][1])
    cams, base_dirs, eyes, ups = cu.create_cameras(nv, d, 64, distribution=dist, device=torch.device('cpu'))
    assert np.array_equal(np.asarray(eyes, np.float64), g[tag + '_eyes'])
    assert np.array_equal(base_dirs.numpy(), g[tag + '_base_dirs']) and np.array_equal(ups.numpy(), g[tag + '_up_dirs'])
    assert len(cams) == len(g[tag + '_eyes']) == len(g[tag + '_fov'])
    for c, fov in zip(cams, g[tag + '_fov']):
        assert abs(float(c.params[12]) - 1.0 / math.tan(fov / 2.0)) < 1e-6 and float(c.params[12]) == float(c.params[13])


def test_c0_camera_distribution_errors():
    import pointdreamer_amd.camera_utils as cu
    with pytest.raises(ValueError):
        cu.create_cameras(8, 1.6, 64, distribution='self_defined', device=torch.device('cpu'))      # the reference knows 6 or 20 views only
    with pytest.raises(ValueError):
        cu.create_cameras(8, 1.6, 64, distribution='spiral', device=torch.device('cpu'))


def test_shapes_uniform_and_stack_host_logic():
    """pointdreamer_amd/shapes.py (round 4): which batches take the one-launch-per-stage route (equal tensor shapes throughout) and how
    their inputs are stacked ([S, ...], int32 faces, int64 face ids, the leading singleton of the atlas maps dropped)."""
    import torch
    from pointdreamer_amd import shapes as shp

    def mk(n, vn=10, f=7, a=16, seed=0):
        g = torch.Generator().manual_seed(seed)
        return dict(coords=torch.rand((n, 3), generator=g), colors=torch.rand((n, 3), generator=g), vertices=torch.rand((vn, 3), generator=g),
                    faces=torch.randint(0, vn, (f, 3), generator=g), f_normals=torch.rand((f, 3), generator=g),
                    xatlas=dict(gb_pos=torch.rand((1, a, a, 3), generator=g), mask=torch.rand((1, a, a, 1), generator=g) > 0.5,
                                per_atlas_pixel_face_id=torch.randint(-1, f, (1, a, a), generator=g)))
    a, b, c = mk(50, seed=1), mk(50, seed=2), mk(51, seed=3)
    assert shp.uniform([a, b]) and not shp.uniform([a, c]) and not shp.uniform([a, mk(50, a=32)])
    # (round 6) meshes of different size stack too: vertices / faces are padded (vertex 0, degenerate faces), `ragged` says so
    g8 = mk(50, vn=12, f=8, seed=7)
    assert shp.uniform([a, g8]) and shp.ragged([a, g8]) and not shp.ragged([a, b])
    sr = shp.stack([a, g8])
    assert sr['vertices'].shape == (2, 12, 3) and sr['faces'].shape == (2, 8, 3) and sr['f_normals'].shape == (2, 8, 3)
    assert torch.equal(sr['vertices'][0, 10:], a['vertices'][:1].expand(2, 3)) and torch.equal(sr['faces'][0, 7], torch.zeros(3, dtype=torch.int32))
    assert torch.equal(sr['faces'][1], g8['faces'].int()) and torch.equal(sr['f_normals'][0, 7], torch.zeros(3))
    #  equal shapes are not enough: one device throughout, atlas maps with their leading singleton dimension
    d = mk(50, seed=4); d['colors'] = d['colors'].to('meta')
    e = mk(50, seed=5); e['xatlas'] = dict(e['xatlas'], mask=e['xatlas']['mask'][0])
    f_ = mk(50, seed=6); f_['xatlas'] = dict(f_['xatlas'], gb_pos=f_['xatlas']['gb_pos'][0].numpy())
    assert not shp.uniform([a, d]) and not shp.uniform([d, d]) and not shp.uniform([a, e]) and not shp.uniform([e, e]) and not shp.uniform([a, f_])
    assert not shp.uniform([])
    st = shp.stack([a, b])
    assert st['coords'].shape == (2, 50, 3) and st['faces'].dtype == torch.int32 and st['faces'].shape == (2, 7, 3)
    assert st['gb_pos'].shape == (2, 16, 16, 3) and st['mask'].shape == (2, 16, 16, 1) and st['face_id'].dtype == torch.int64
    assert torch.equal(st['coords'][1], b['coords']) and torch.equal(st['face_id'][0], a['xatlas']['per_atlas_pixel_face_id'][0])
    assert all(t.is_contiguous() for t in st.values())
