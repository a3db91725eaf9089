"""End-to-end demo driver on the GPU: reference CLI/config surface in, reference output tree out."""
import os
import numpy as np
import PIL.Image
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_cloud(path, n=20000, seed=3):
    from pointdreamer_amd import synthetic, io_utils
    xyz, rgb = synthetic.sphere_points(n, seed=seed)
    io_utils.save_colored_pc_ply(xyz * 1.7 + 0.3, rgb, path)          # off-centre / scaled: the driver normalises it


@pytest.mark.parametrize("config,extra", [("nearest.yaml", []), ("default.yaml", ["--allow_random_weights"])])
def test_demo_cli_outputs(tmp_path, config, extra):
    from pointdreamer_amd import demo
    pc = str(tmp_path / 'ball.ply')
    _write_cloud(pc)
    over = ["--set", f"output_path={tmp_path / 'out'}", "xatlas_texture_res=512", "point_validation_by_o3d=False"]
    if config == "default.yaml":
        import pointdreamer_amd.ddnm_inpainting as di
        orig = di.Inpainter.inpaint_views
        di.Inpainter.inpaint_views = lambda self, a, b, **k: orig(self, a, b, n_steps=2)      # 2 DDNM steps keep the test short
    try:
        outs = demo.main(["--config", os.path.join(ROOT, "configs", config), "--pc_file", pc] + extra + over)
    finally:
        if config == "default.yaml":
            di.Inpainter.inpaint_views = orig
    out = outs[0]
    assert os.path.basename(out) == "ball_" + config.split('.')[0]
    for f in ["config.yaml", "input_pc.ply", "models/model_normalized.obj", "models/model_normalized.mtl",
              "models/model_normalized.png", "others/atlas_wo_background.png"] + \
             [f"others/{k}_{s}.png" for k in range(8) for s in ("sparse", "mask0", "mask2", "inpainted")]:
        assert os.path.exists(os.path.join(out, f)), f
    atlas = np.array(PIL.Image.open(os.path.join(out, "models/model_normalized.png")))
    assert atlas.shape == (512, 512, 3) and atlas.std() > 5
    rgba = np.array(PIL.Image.open(os.path.join(out, "others/atlas_wo_background.png")))
    assert rgba.shape[2] == 4 and set(np.unique(rgba[..., 3])) <= {0, 255}
    mode = PIL.Image.open(os.path.join(out, "others/0_inpainted.png")).mode
    assert mode == ("RGBA" if config == "default.yaml" else "RGB")          # ours_utils.py:924-928 vs :939-941
    if config == "nearest.yaml":
        # colours come from the cloud: the nearest atlas must correlate with the analytic colour field
        sp = np.array(PIL.Image.open(os.path.join(out, "others/0_sparse.png")))
        assert sp.shape == (256, 256, 4) and (sp[..., 3] > 0).mean() > 0.05


def test_demo_with_supplied_mesh_obj_builds_and_caches_the_atlas(tmp_path):
    """demo.py:391-399 hook: `<pc>_untextured_mesh.obj` next to the PLY.  The OBJ carries vt / f v/vt records, so the atlas
    producer (8f-3) rasterises the UV atlas and caches geo/xatlas_<res>.pth in the reference's wire format; a second run loads it."""
    import torch
    from pointdreamer_amd import demo, synthetic, io_utils
    pc = str(tmp_path / 'ball.ply')
    _write_cloud(pc)
    verts, faces, _ = synthetic.uv_sphere(24, 48)
    uvs, fuv = synthetic.uv_sphere_uvs(24, 48, 512, gutter=2)
    io_utils.savemeshtes2(verts, uvs, faces, fuv, str(tmp_path / 'ball_untextured_mesh.obj'))
    args = ["--config", os.path.join(ROOT, "configs", "nearest.yaml"), "--pc_file", pc, "--set", f"output_path={tmp_path / 'out'}",
            "xatlas_texture_res=512", "point_validation_by_o3d=False"]
    out = demo.main(args)[0]
    cache = os.path.join(out, "geo", "xatlas_512.pth")
    assert os.path.exists(cache)
    d = torch.load(cache)
    assert set(d) == {"uvs", "mesh_tex_idx", "gb_pos", "mask", "per_atlas_pixel_face_id"}
    assert tuple(d["gb_pos"].shape) == (1, 512, 512, 3) and tuple(d["mask"].shape) == (1, 512, 512, 1)
    assert 0.5 < d["mask"].float().mean() < 1.0
    a1 = np.array(PIL.Image.open(os.path.join(out, "models/model_normalized.png")))
    assert a1.std() > 5
    out2 = demo.main(args)[0]                                   # second run: cached dict AND the {i}_inpainted.png of the first run
    a2 = np.array(PIL.Image.open(os.path.join(out2, "models/model_normalized.png")))   # (demo.py:138-147: loaded back in 8 bits)
    assert (np.abs(a1.astype(int) - a2.astype(int)) <= 3).mean() > 0.98
    for k in range(8):
        os.remove(os.path.join(out, "others", f"{k}_inpainted.png"))
    out3 = demo.main(args)[0]                                   # cached dict only: bit-identical to the first run
    a3 = np.array(PIL.Image.open(os.path.join(out3, "models/model_normalized.png")))
    assert np.array_equal(a1, a3)


def test_demo_directory_run_batches_shapes_and_matches_one_at_a_time(tmp_path):
    """`--pc_file <dir>`: clouds are textured in groups (views of a group in one inpainter batch); same files and the same atlas
    as the one-at-a-time run ('nearest' config: deterministic)."""
    from pointdreamer_amd import demo
    d = tmp_path / 'clouds'
    d.mkdir()
    for k in range(3):
        _write_cloud(str(d / f'c{k}.ply'), seed=20 + k)
    cfgf = os.path.join(ROOT, "configs", "nearest.yaml")
    over = ["xatlas_texture_res=512"]
    outs_b = demo.main(["--config", cfgf, "--pc_file", str(d), "--batch_shapes", "2", "--set", f"output_path={tmp_path / 'b'}"] + over)
    outs_1 = demo.main(["--config", cfgf, "--pc_file", str(d), "--batch_shapes", "1", "--set", f"output_path={tmp_path / 'o'}"] + over)
    assert len(outs_b) == len(outs_1) == 3
    for ob, o1 in zip(outs_b, outs_1):
        assert os.path.basename(ob) == os.path.basename(o1)
        for f in ["config.yaml", "input_pc.ply", "models/model_normalized.obj", "models/model_normalized.mtl", "models/model_normalized.png",
                  "others/atlas_wo_background.png"] + [f"others/{k}_{s}.png" for k in range(8) for s in ("sparse", "mask0", "mask2", "inpainted")]:
            assert os.path.exists(os.path.join(ob, f)), f
        a = np.array(PIL.Image.open(os.path.join(ob, "models/model_normalized.png")))
        b = np.array(PIL.Image.open(os.path.join(o1, "models/model_normalized.png")))
        assert np.array_equal(a, b)
