"""CPU tests of the I/O edges (row O1 + PLY input): formats must match the reference's writers/readers
(utils/utils_2d.py:351-399, models/get3d/get3d_utils/utils_3d.py:27-64, utils/other_utils.py:122-163)."""
import os
import numpy as np
import PIL.Image
from pointdreamer_amd import io_utils


def test_ply_roundtrip_binary_le(tmp_path):
    rng = np.random.default_rng(0)
    xyz = rng.uniform(-1, 1, (1000, 3)).astype(np.float32)
    rgb = rng.uniform(0, 1, (1000, 3)).astype(np.float32)
    p = str(tmp_path / 'a.ply')
    io_utils.save_colored_pc_ply(xyz, rgb, p)
    # 30 000-vertex demo PLYs are 450 179 bytes: header (179) + 15 bytes per vertex
    assert os.path.getsize(p) - 1000 * 15 == len(open(p, 'rb').read().split(b'end_header\n')[0]) + len(b'end_header\n')
    x2, c2 = io_utils.read_ply_xyzrgb(p)
    assert np.array_equal(x2, xyz)
    assert np.array_equal(c2, (rgb * 255).astype(np.uint8))


def test_png_truncation_not_rounding(tmp_path):
    img = np.zeros((3, 2, 2), np.float32)
    img[0, 0, 0] = 0.999          # 254.745 -> 254 (truncation), not 255
    img[1, 0, 1] = 1.7            # clipped to 255
    img[2, 1, 0] = -0.2           # clipped to 0
    keep = img.copy()
    p = str(tmp_path / 'x.png')
    io_utils.save_CHW_RGB_img(img, p)
    assert np.array_equal(img, keep), "writer must not scale the caller's array in place (the reference does)"
    a = np.array(PIL.Image.open(p))
    assert a[0, 0, 0] == 254 and a[0, 1, 1] == 255 and a[1, 0, 2] == 0
    back = io_utils.load_CHW_RGB_img(p)
    assert back.shape == (3, 2, 2) and abs(float(back[0, 0, 0]) - 254 / 255) < 1e-7
    rgba = np.concatenate([img, np.ones((1, 2, 2), np.float32)], 0)
    io_utils.save_CHW_RGBA_img(rgba, str(tmp_path / 'y.png'))
    assert PIL.Image.open(str(tmp_path / 'y.png')).mode == 'RGBA'


def test_obj_mtl_text_format(tmp_path):
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.5]], np.float32)
    vt = np.array([[0.1, 0.2], [0.3, 0.4], [0.5, 0.6]], np.float32)
    f = np.array([[0, 1, 2]])
    p = str(tmp_path / 'model_normalized.obj')
    io_utils.savemeshtes2(v, vt, f, f, p)
    lines = open(p).read().splitlines()
    assert lines[0] == 'mtllib model_normalized.mtl'
    assert lines[1] == 'v 0.000000 0.000000 0.000000' and lines[3] == 'v 0.000000 1.000000 0.500000'
    assert lines[4] == 'vt 0.100000 0.200000'
    assert lines[7] == 'usemtl material_0' and lines[8] == 'f 1/1 2/2 3/3'
    mtl = open(str(tmp_path / 'model_normalized.mtl')).read().splitlines()
    assert mtl == ['newmtl material_0', 'Kd 1 1 1', 'Ka 0 0 0', 'Ks 0.4 0.4 0.4', 'Ns 10', 'illum 2', 'map_Kd model_normalized.png']
    vv, ff = io_utils.load_obj_mesh(p)
    assert np.allclose(vv, v) and np.array_equal(ff, f)
