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
