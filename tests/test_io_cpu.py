"""CPU tests of the I/O edges (row O1 + PLY input): formats must match the reference's writers/readers
(utils/utils_2d.py:351-399, models/get3d/get3d_utils/utils_3d.py:27-64, utils/other_utils.py:122-163)."""
import os
import numpy as np
import PIL.Image
import pytest
import torch
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


def test_native_obj_writer_is_byte_identical_to_python_percent_f(tmp_path):
    """8f-4: the native OBJ writer against the reference's own formatting expression (`'%f' % numpy_float32`)."""
    rng = np.random.default_rng(1)
    v = (rng.standard_normal((500, 3)) * 10 ** rng.uniform(-6, 3, (500, 1))).astype(np.float32)
    vt = rng.uniform(0, 1, (700, 2)).astype(np.float32)
    f = rng.integers(0, 500, (900, 3))
    ft = rng.integers(0, 700, (900, 3))
    p = str(tmp_path / 'm.obj')
    io_utils.savemeshtes2(v, vt, f, ft, p)
    exp = ['mtllib m.mtl\n'] + ['v %f %f %f\n' % (q[0], q[1], q[2]) for q in v] + ['vt %f %f\n' % (q[0], q[1]) for q in vt]
    exp += ['usemtl material_0\n'] + ['f %d/%d %d/%d %d/%d\n' % (a[0], b[0], a[1], b[1], a[2], b[2]) for a, b in zip(f + 1, ft + 1)]
    assert open(p).read() == ''.join(exp)


def test_native_png_encoder_decodes_to_the_same_pixels(tmp_path):
    rng = np.random.default_rng(2)
    for ch, mode in ((3, 'RGB'), (4, 'RGBA')):
        img = rng.uniform(-0.1, 1.1, (ch, 37, 53)).astype(np.float32)
        p = str(tmp_path / f'r{ch}.png')
        (io_utils.save_CHW_RGB_img if ch == 3 else io_utils.save_CHW_RGBA_img)(img, p)
        im = PIL.Image.open(p)
        assert im.mode == mode and im.size == (53, 37)
        assert np.array_equal(np.array(im), (img.transpose(1, 2, 0) * 255).clip(0, 255).astype(np.uint8))


def test_native_ply_reader_ascii_and_extra_properties(tmp_path):
    p = str(tmp_path / 'a.ply')
    with open(p, 'w') as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                "property float nx\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nelement face 0\n"
                "property list uchar int vertex_indices\nend_header\n0 0.5 1 9 255 0 7\n-1.25 2 3 9 1 2 3\n4 5 6.5 9 10 20 30\n")
    xyz, rgb = io_utils.read_ply_xyzrgb(p)
    assert np.array_equal(xyz, np.array([[0, 0.5, 1], [-1.25, 2, 3], [4, 5, 6.5]], np.float32))
    assert np.array_equal(rgb, np.array([[255, 0, 7], [1, 2, 3], [10, 20, 30]], np.uint8))
    # binary with a double-typed coordinate and a leading extra property
    q = str(tmp_path / 'b.ply')
    v = np.zeros(2, dtype=[('s', '<i2'), ('x', '<f8'), ('y', '<f4'), ('z', '<f4'), ('red', 'u1'), ('green', 'u1'), ('blue', 'u1')])
    v['s'], v['x'], v['y'], v['z'], v['red'], v['green'], v['blue'] = [7, 8], [0.25, -3.5], [1, 2], [3, 4], [9, 8], [7, 6], [5, 4]
    with open(q, 'wb') as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty short s\nproperty double x\nproperty float y\n"
                b"property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" + v.tobytes())
    xyz, rgb = io_utils.read_ply_xyzrgb(q)
    assert np.array_equal(xyz, np.array([[0.25, 1, 3], [-3.5, 2, 4]], np.float32)) and np.array_equal(rgb, np.array([[9, 7, 5], [8, 6, 4]], np.uint8))


def test_native_png_parallel_strips_decode_to_the_same_pixels(tmp_path):
    """Images of 512 KiB and more are deflated in parallel strips (pigz-style concatenation): still one valid zlib stream."""
    import PIL.Image
    rng = np.random.default_rng(5)
    for ch, h, w in ((3, 1024, 1024), (4, 700, 333), (3, 419, 419)):          # 419*420*3 = just above the threshold
        a = rng.random((ch, h, w)).astype(np.float32)
        a[:, : h // 3] = 0.25                                                   # compressible and incompressible strips
        f = str(tmp_path / f"big{ch}_{h}.png")
        (io_utils.save_CHW_RGB_img if ch == 3 else io_utils.save_CHW_RGBA_img)(torch.from_numpy(a), f)
        want = (a.transpose(1, 2, 0) * 255.0).clip(0, 255).astype(np.uint8)
        assert np.array_equal(np.array(PIL.Image.open(f)), want)


def test_deferred_writes_land_on_flush_and_report_errors(tmp_path):
    import PIL.Image
    rng = np.random.default_rng(6)
    imgs = [rng.random((3, 64, 80)).astype(np.float32) for _ in range(12)]
    io_utils.set_async(True, workers=4)
    try:
        for i, a in enumerate(imgs):
            io_utils.save_CHW_RGB_img(torch.from_numpy(a), str(tmp_path / f"{i}.png"))
        v = rng.random((5, 3)); uv = rng.random((5, 2)); f = np.array([[0, 1, 2], [2, 3, 4]])
        io_utils.savemeshtes2(v, uv, f, f, str(tmp_path / "m.obj"))
        io_utils.flush()
        for i, a in enumerate(imgs):
            assert np.array_equal(np.array(PIL.Image.open(tmp_path / f"{i}.png")), (a.transpose(1, 2, 0) * 255.0).clip(0, 255).astype(np.uint8))
        assert (tmp_path / "m.obj").exists() and (tmp_path / "model_normalized.mtl").exists()
        io_utils.save_CHW_RGB_img(torch.from_numpy(imgs[0]), str(tmp_path / "no_such_dir" / "x.png"))
        with pytest.raises(Exception):
            io_utils.flush()
    finally:
        io_utils.set_async(False)
    # back to synchronous: the file exists on return
    io_utils.save_CHW_RGB_img(torch.from_numpy(imgs[0]), str(tmp_path / "sync.png"))
    assert (tmp_path / "sync.png").exists()


def test_o1_writers_vs_reference_golden(tmp_path):
    """Row O1 pinned: bytes / pixels written by the reference's own savemeshtes2 (utils_3d.py:27-64), save_CHW_RGB(A)_img
    (utils_2d.py:351-381) and demo.save_textured_mesh (demo.py:264-307), recorded by tools/gen_golden_r2.py."""
    from conftest import load_golden
    from pointdreamer_amd import demo
    g = load_golden('o1_writers.npz')
    p = str(tmp_path / 'model_normalized.obj')
    io_utils.savemeshtes2(g['vertices'], g['uvs'], g['faces'], g['face_uv_idx'], p)
    assert open(p, 'rb').read() == g['ref_obj'].tobytes()
    assert open(str(tmp_path / 'model_normalized.mtl'), 'rb').read() == g['ref_mtl'].tobytes()
    io_utils.save_CHW_RGB_img(g['img_rgb'].copy(), str(tmp_path / 'rgb.png'))
    io_utils.save_CHW_RGBA_img(g['img_rgba'].copy(), str(tmp_path / 'rgba.png'))
    a, b = PIL.Image.open(str(tmp_path / 'rgb.png')), PIL.Image.open(str(tmp_path / 'rgba.png'))
    assert a.mode == 'RGB' and b.mode == 'RGBA'
    assert np.array_equal(np.array(a), g['ref_rgb_pixels']) and np.array_equal(np.array(b), g['ref_rgba_pixels'])
    assert np.array_equal(io_utils.load_CHW_RGB_img(str(tmp_path / 'rgba.png')).numpy(), g['ref_loaded_from_rgba'])
    for d in ('models', 'others'):
        os.makedirs(str(tmp_path / d))
    T = torch.from_numpy
    demo.save_textured_mesh(T(g['vertices']), T(g['uvs']), T(g['faces']), T(g['face_uv_idx']), T(g['atlas']), T(g['atlas_mask']), str(tmp_path))
    assert open(str(tmp_path / 'models' / 'model_normalized.obj'), 'rb').read() == g['ref_obj2'].tobytes()
    assert np.array_equal(np.array(PIL.Image.open(str(tmp_path / 'models' / 'model_normalized.png'))), g['ref_atlas_png'])
    assert np.array_equal(np.array(PIL.Image.open(str(tmp_path / 'others' / 'atlas_wo_background.png'))), g['ref_atlas_rgba_png'])


def test_reference_yaml_configs_load_verbatim(tmp_path):
    """The reference's five configs/*.yaml (their parsed key -> value tables, tests/golden/reference_configs.json) go through the
    driver's config loader unchanged: path keys are honoured, upstream keys (dataset / POCO / SPR / evaluation) are accepted and
    ignored, a mistyped key is an error.  The shipped configs carry the same values for every key the path reads."""
    import json
    import yaml
    from pointdreamer_amd import demo
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, 'tests', 'golden', 'reference_configs.json')))
    assert sorted(ref) == ['default.yaml', 'geo_by_SPR.yaml', 'nearest.yaml', 'noisy.yaml', 'wo_NBF.yaml']
    for name, table in ref.items():
        p = str(tmp_path / name)
        yaml.safe_dump(table, open(p, 'w'))
        cfg = demo.load_config(p)
        for k in demo.SUPPORTED_KEYS:
            if k in table and k != 'optimize_from':
                assert cfg[k] == table[k], (name, k)
        assert cfg.optimize_from in (None, 'scratch', 'naive', 'ours')
        kw = demo._pipeline_kwargs(cfg)
        assert 'geo_from' not in kw and 'exp_name' not in kw and kw['view_num'] == table['view_num']
        ours = demo.load_config(os.path.join(root, 'configs', name))            # the shipped twin
        for k in demo.SUPPORTED_KEYS:
            if k in table:
                assert ours[k] == cfg[k], (name, k, ours[k], cfg[k])
    bad = dict(ref['nearest.yaml'], edge_dilate_kernel=[21])
    yaml.safe_dump(bad, open(str(tmp_path / 'bad.yaml'), 'w'))
    with pytest.raises(KeyError):
        demo.load_config(str(tmp_path / 'bad.yaml'))


def test_host_thread_budget_is_shared_between_the_ranks_of_a_node(monkeypatch):
    """bench.py / demo.py cap torch's intra-op pool and the writer pool by cpus_per_rank(): the CPUs the cgroup quota and the affinity
    mask grant (the GPU boxes show 256 logical CPUs and grant 16), divided by torchrun's LOCAL_WORLD_SIZE (one process per GPU)."""
    n = io_utils.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    monkeypatch.delenv('LOCAL_WORLD_SIZE', raising=False)
    assert io_utils.cpus_per_rank() == n
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    assert io_utils.cpus_per_rank() == max(1, n // 8)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '100000')
    assert io_utils.cpus_per_rank() == 1
