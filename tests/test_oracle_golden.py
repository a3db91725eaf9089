"""Oracle (oracle/) vs golden vectors produced by the imported reference (tools/gen_golden.py)."""
import numpy as np
import pytest
from conftest import load_golden
from oracle import camera as ocam, project as oproj, sparse as osparse, inpaint as oinp, nbf as onbf
from oracle import unproject as ounp


@pytest.mark.parametrize("name", ["proj_sparse_dense.npz", "proj_sparse_rescale.npz", "proj_sparse_ps2.npz",
                                  "proj_sparse_scale_near1.npz"])
def test_p3_depth_visibility_bit_exact(name):
    g = load_golden(name)
    vis, pix = oproj.point_validation_by_depth(int(g['cam_res']), g['point_uvs'], g['point_depths'],
                                               g['mesh_depths'], offset=0.0001)
    assert np.array_equal(vis, g['ref_visibility'])
    assert np.array_equal(pix, g['ref_point_pixels_R'])


@pytest.mark.parametrize("name", ["proj_sparse_dense.npz", "proj_sparse_rescale.npz", "proj_sparse_ps2.npz",
                                  "proj_sparse_scale_near1.npz"])
def test_p4_p6_sparse_images_bit_exact(name):
    g = load_golden(name)
    V = g['point_uvs'].shape[0]
    sparse, m0, m2, sf = osparse.get_sparse_images(g['point_pixels_r'], g['colors'], g['ref_visibility'],
                                                   g['hard_masks_r'], V, int(g['res']), int(g['point_size']),
                                                   int(g['edge_point_size']), 0.82)
    assert np.array_equal(m0, g['ref_mask0'])
    assert np.array_equal(m2, g['ref_mask2'])
    assert np.array_equal(sf, g['ref_scale_factors'])
    assert np.array_equal(sparse, g['ref_sparse'])          # colours are exact copies of inputs
    if name == "proj_sparse_rescale.npz":
        assert (sf < 1).any(), "fixture must exercise the mask_ratio > 0.82 branch"


def test_p4_degenerate_view_is_all_background():
    g = load_golden("proj_sparse_dense.npz")
    r = int(g['res'])
    none_valid = np.zeros_like(g['ref_visibility'][0])
    s, m0, m2, ratio, sf = osparse.get_one_sparse_img(g['point_pixels_r'][0], g['colors'], none_valid,
                                                      g['hard_masks_r'][0], r, 1, 1)
    assert sf == 1 and not s.any() and np.array_equal(m2, 1 - m0)


@pytest.mark.parametrize("name", ["nearest_dense.npz", "nearest_rescale.npz"])
def test_i0_nearest_vs_reference_scipy(name):
    g = load_golden(name)
    for v in range(g['sparse'].shape[0]):
        img, m2 = g['sparse'][v], g['mask2'][v]
        out = oinp.nearest_inpaint(img, m2)
        ref = g['ref_inpainted'][v].astype(np.float32)
        sites = m2[0].astype(bool)
        rr, cc = oinp.nearest_site_index(sites)
        # minimal-distance property everywhere (brute force), and the build's tie rule
        H, W = sites.shape
        sr, scol = np.nonzero(sites)
        qi, qj = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
        d2 = (qi.reshape(-1, 1) - sr[None]) ** 2 + (qj.reshape(-1, 1) - scol[None]) ** 2
        dmin = d2.min(1)
        chosen = (qi.reshape(-1) - rr.reshape(-1)) ** 2 + (qj.reshape(-1) - cc.reshape(-1)) ** 2
        assert np.array_equal(chosen, dmin)
        assert sites[rr, cc].all()
        key = np.where(d2 == dmin[:, None], sr[None] * W + scol[None], 1 << 40).min(1)
        assert np.array_equal(key, rr.reshape(-1) * W + cc.reshape(-1))
        # identical to the reference wherever the nearest site is unique
        unique = ((d2 == dmin[:, None]).sum(1) == 1).reshape(H, W)
        assert unique.mean() > 0.3
        assert np.array_equal(out[:, unique], ref[:, unique])
        # at tie pixels the reference's pick is also at minimal distance: its colour is one of the tied sites'
        assert np.array_equal(out[:, sites], img[:, sites])


def test_n1_n2_integer_identities():
    g = load_golden("unproject_k21.npz")
    e = onbf.scharr_edges_binary(g['n1_in'])
    assert np.array_equal(e, g['n1_ref_edges_gt125'][:, 0])
    assert np.array_equal(e, g['n1_ref_edges_gt126_5'][:, 0])
    d = onbf.dilate_binary(e, 7)
    assert np.array_equal(d, g['n2_ref_dilated7'])


def _run_unproject(g):
    cams = [ocam.Camera(p, int(g['cam_res'])) for p in g['cam_params']]
    return ounp.unproject(g['inpainted'], g['f_normals'], int(g['res']), cams, int(g['cam_res']), g['base_dirs'],
                          g['gb_pos'], g['mask'], g['face_id'], g['uv_centers'], g['uv_scales'], float(g['padding']),
                          g['scale_factors'], g['mesh_depths'], [int(k) for k in g['kernels']], bool(g['complete']))


@pytest.mark.parametrize("name", ["unproject_k21.npz", "unproject_k21_complete.npz", "unproject_k0.npz",
                                  "unproject_multi.npz"])
def test_uq_unproject_vs_reference(name):
    g = load_golden(name)
    o = _run_unproject(g)
    assert np.array_equal(o['points_atlas_pixel_coord'], g['ref_coords'])
    assert np.array_equal(o['points'], g['ref_points'])
    assert np.array_equal(o['shrinked'], g['ref_shrinked'])
    # view ids: bit-identical except where the two best candidate similarities are within 1e-6
    # (the reference's sgemm accumulation order is not ours); expect zero or a handful.
    diff = o['point_view_ids'] != g['ref_view_ids']
    if diff.any():
        sim = np.sort(o['sim'][diff], 1)
        assert (sim[:, -1] - sim[:, -2] < 1e-6).all()
        assert diff.mean() < 1e-3
    same = ~diff
    coords = g['ref_coords'][same]
    assert np.array_equal(o['atlas_img'][coords[:, 0], coords[:, 1]], g['ref_atlas'][coords[:, 0], coords[:, 1]])
    if not diff.any():
        assert np.array_equal(o['atlas_img'], g['ref_atlas'])
        assert np.array_equal(o['atlas_painted_mask'], g['ref_painted'])
    assert (o['point_view_ids'] >= 0).any()
    if not bool(g['complete']):
        assert (o['point_view_ids'] == -100).any(), "fixture should contain unseen texels"


def test_uq5_dilate_atlas_vs_reference():
    g = load_golden("unproject_k21_complete.npz")
    out = oinp.dilate_atlas(g['ref_atlas'], g['mask'])
    m = g['mask'][0, :, :, 0]
    assert np.array_equal(out[m], g['ref_atlas'][m])
    rr, cc = oinp.nearest_site_index(m)
    # reference picks a site at the same (minimal) distance everywhere
    ref = g['ref_dilated']
    same = (out == ref).all(-1)
    assert same.mean() > 0.6
    H, W = m.shape
    qi, qj = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    d_mine = (qi - rr) ** 2 + (qj - cc) ** 2
    # every reference colour at a differing pixel must equal the colour of SOME site at distance d_mine
    bad = np.argwhere(~same)[:200]
    sr, sc = np.nonzero(m)
    for (i, j) in bad:
        d2 = (sr - i) ** 2 + (sc - j) ** 2
        tied = d2 == d_mine[i, j]
        assert d2.min() == d_mine[i, j]
        assert (g['ref_atlas'][sr[tied], sc[tied]] == ref[i, j]).all(-1).any()


@pytest.mark.parametrize("name", ["neighbor_small.npz", "neighbor_seam.npz"])
def test_8f2_neighbor_completion_vs_reference(name):
    """paint_invisible_areas_by_neighbors (unproject.py:93-196) + subdivide_with_uv (mesh_utils.py:7-114): the oracle against the
    imported reference's outputs (tools/gen_golden_neighbor.py).  Subdivided mesh bit-exact; vertex colours / the atlas handed
    to the final fill to 1e-6 (the reference's dense sgemm order is not ours); the nearest-filled atlas identical wherever the
    nearest painted texel is unique (scipy's KD-tree picks an arbitrary one among ties)."""
    from oracle import neighbor as onb
    from pointdreamer_amd import mesh_utils as mu
    g = load_golden(name)
    o = onb.paint_invisible_areas_by_neighbors(g['vertices'], g['faces'], g['uvs'], g['face_uv_idx'], g['to_inpaint_face_id'],
                                               g['atlas'], g['painted'], return_intermediates=True)
    assert np.array_equal(o['vertices'], g['ref_sub_vertices']) and np.array_equal(o['faces'], g['ref_sub_faces'])
    assert np.abs(o['vert_colors'] - g['ref_vert_colors']).max() <= 1e-6
    assert np.abs(o['atlas_before_fill'] - g['ref_atlas_before_fill']).max() <= 1e-6
    assert np.array_equal(o['mask_before_fill'], g['ref_mask_before_fill'] > 0)
    m = o['mask_before_fill']
    assert np.abs(o['atlas'][m] - g['ref_atlas'][m]).max() <= 1e-6
    same = (np.abs(o['atlas'] - g['ref_atlas']).max(-1) <= 1e-6)
    assert same.mean() > 0.95
    # the product's host-side mesh helpers (pointdreamer_amd/mesh_utils.py) give the same numbering
    v, f, u, fu = g['vertices'], g['faces'], g['uvs'], g['face_uv_idx']
    for _ in range(2):
        v, f, u, fu = mu.subdivide_with_uv(v, f, fu, u, face_index=g['to_inpaint_face_id'])
    assert np.array_equal(v, g['ref_sub_vertices']) and np.array_equal(f, g['ref_sub_faces'])
    assert np.array_equal(u, o['uvs']) and np.array_equal(fu, o['face_uv_idx'])
    assert np.array_equal(mu.vertex_uv_table(len(v), f, fu, u), onb.vertex_uvs(len(v), f, fu, u))
    r1, c1 = mu.neighbour_csr(len(v), f)
    r2, c2 = onb.adjacency_csr(len(v), f)
    assert np.array_equal(r1, r2) and np.array_equal(c1, c2)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_p1_crop_arithmetic_vs_reference(tag):
    """Row P1 pinned: the reference's own get_rendered_hard_mask_and_face_idx_batch (ours_utils.py:93-150, run by
    tools/gen_golden_r2.py with the oracle camera and raster as the two third-party stand-ins) against the oracle."""
    g = load_golden('p1_crop.npz')
    G = lambda k: g[f'{tag}_{k}']
    cams = [ocam.Camera(p, int(G('cam_res'))) for p in G('cam_params')]
    o = oproj.project_batch(cams, G('vertices'), G('points'), bool(G('rescale')), float(G('padding')))
    assert np.array_equal(o['pos'], G('ref_pos'))
    assert np.array_equal(o['vertice_uvs'], G('ref_vertice_uvs'))
    assert np.array_equal(o['point_uvs'], G('ref_point_uvs'))
    assert np.array_equal(o['point_depths'], G('ref_point_depths'))
    assert np.array_equal(np.asarray(o['uv_centers'], np.float32), G('ref_uv_centers'))
    assert np.array_equal(np.asarray(o['uv_scales'], np.float32), G('ref_uv_scales'))
    assert np.float32(o['padding']) == G('ref_padding')
    hard, fid, depth = oproj.rasterize(o['pos'], G('faces'), int(G('cam_res')))
    assert np.array_equal(hard, G('ref_hard')) and np.array_equal(fid, G('ref_face_idx')) and np.array_equal(depth, G('ref_depth'))


def test_o1_shrink_triptychs_vs_reference_pngs():
    """unproject.py:459-474 with the reference's own cat_images / save_CHW_RGB_img (decoded PNG pixels)."""
    g = load_golden('triptych.npz')
    ks = [int(k) for k in g['kernels']]
    assert np.array_equal(onbf.shrink_visibility(g['mask'], g['vis'], ks), g['ref_shrinked'])
    assert np.array_equal(onbf.shrink_triptychs(g['mask'], g['vis'], ks), g['ref_pngs'])


def test_c0_camera_hand_computed_cases():
    """kaolin is absent (C0 parity unpinned): pin the oracle's restatement of its documented pinhole / OpenGL-NDC convention
    (kaolin 0.15 `Camera.from_args(eye, at, up, fov=pi/4, near=1e-2, far=1e2)`) with values computed by hand, independent of the
    code under test: x_ndc = f x_c / -z_c with f = 1 / tan(fov / 2), z_ndc = -1 at the near plane, +1 at the far plane,
    increasing with distance; row/column orientation: +x right, +y up in NDC."""
    import math
    cam = ocam.Camera(ocam.camera_params(np.array([0.0, 0.0, 1.6]), np.zeros(3), np.array([0.0, 1.0, 0.0])), 512)
    f = 1.0 / math.tan(math.pi / 8)
    pts = np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.25, 0], [0, 0, 1.6 - 1e-2], [0, 0, 1.6 - 1e2], [0.3, -0.2, 0.4]], np.float32)
    got = cam.transform(pts).astype(np.float64)
    n, fa = 1e-2, 1e2
    zndc = lambda d: ((fa + n) / (fa - n) * d - 2 * fa * n / (fa - n)) / d          # d = distance along the view axis
    want = np.array([[0, 0, zndc(1.6)], [f * 0.5 / 1.6, 0, zndc(1.6)], [0, f * 0.25 / 1.6, zndc(1.6)], [0, 0, -1.0], [0, 0, 1.0],
                     [f * 0.3 / 1.2, f * -0.2 / 1.2, zndc(1.2)]])
    assert np.allclose(got, want, atol=2e-6), (got, want)
    assert abs(zndc(1.6) - 0.98769877) < 1e-6                                         # (1.0002 * 1.6 - 0.020002) / 1.6
    # a camera off the axis: the look-at rotation keeps the target in the image centre and `up` upright
    eye = np.array([1.0, 0.8, -0.6]) * (1.6 / np.linalg.norm([1.0, 0.8, -0.6]))
    cam2 = ocam.Camera(ocam.camera_params(eye, np.zeros(3), ocam.calculate_up_vector(eye, np.zeros(3))), 512)
    c = cam2.transform(np.array([[0, 0, 0], [0, 0.1, 0]], np.float32))
    assert np.allclose(c[0, :2], 0, atol=1e-6) and abs(c[1, 0]) < 1e-6 and c[1, 1] > 0


def test_p2_raster_hand_computed_cases():
    """nvdiffrast is absent (P2 parity unpinned): pin the documented contract with cases small enough to do by hand --
    pixel (row i, col j) samples NDC ((j + 0.5) / R * 2 - 1, (i + 0.5) / R * 2 - 1) (row 0 = y -1), a pixel centre exactly on a
    shared edge belongs to exactly one of the two triangles, depth = barycentric z, nearest z wins, output face id / depth 0
    where empty."""
    R = 4
    # one triangle covering the lower-left half of the square [-1, 1]^2 (vertices at NDC corners), z = 0.25 everywhere
    pos = np.array([[[-1, -1, 0.25, 1], [1, -1, 0.25, 1], [-1, 1, 0.25, 1], [1, 1, 0.75, 1]]], np.float32)
    hard, fid, depth = oproj.rasterize(pos, np.array([[0, 1, 2]]), R)
    cx = (np.arange(R) + 0.5) / R * 2 - 1
    inside = (cx[None, :] + cx[:, None]) < 0                                        # x + y < 0 strictly below the diagonal
    on = np.isclose(cx[None, :] + cx[:, None], 0)
    assert np.array_equal(hard[0] & ~on, inside & ~on)                               # (the diagonal's own pixels: tie rule, next)
    assert np.allclose(depth[0][hard[0]], 0.25) and (depth[0][~hard[0]] == 0).all() and (fid[0][~hard[0]] == -1).all()
    # both halves: every pixel covered exactly once, the diagonal's pixel centres go to one triangle only (watertight)
    hard2, fid2, depth2 = oproj.rasterize(pos, np.array([[0, 1, 2], [1, 3, 2]]), R)
    assert hard2.all() and set(np.unique(fid2)) == {0, 1}
    assert np.array_equal(fid2[0][inside & ~on], np.zeros(int((inside & ~on).sum()), fid2.dtype))
    # second triangle's depth is the plane through z = 0.25, 0.75, 0.25 at its corners: z = 0.25 + 0.25 (x + y) on x + y >= 0
    yy, xx = np.meshgrid(cx, cx, indexing='ij')
    m = fid2[0] == 1
    assert np.allclose(depth2[0][m], 0.25 + 0.25 * (xx + yy)[m], atol=1e-6)
    # a nearer triangle in front wins the depth test
    pos3 = np.concatenate([pos[0], np.array([[-1, -1, -0.5, 1], [1, -1, -0.5, 1], [-1, 1, -0.5, 1]], np.float32)])[None]
    _, fid3, depth3 = oproj.rasterize(pos3, np.array([[0, 1, 2], [1, 3, 2], [4, 5, 6]]), R)
    assert (fid3[0][inside & ~on] == 2).all() and np.allclose(depth3[0][inside & ~on], -0.5)


@pytest.mark.parametrize("name,tol", [("optimize_64_3_1.npz", 1e-6), ("optimize_128_3_0.npz", 1e-6), ("optimize_64_100_1.npz", None)])
def test_optimize_color_oracle_vs_reference_loop(name, tol):
    """SURVEY 8f-1 / VERDICT r5 item 2: oracle/optimize.py against the atlas the REFERENCE's own optimize_color returned
    (pointdreamer/ours_utils.py:1583-1785 run by tools/gen_golden_r2.py gen_optimize: Adam 5e-2, StepLR(15, 0.5), f64 bilinear lookup, L1 masked by
    foreground and shrunk visibility, its hard-wired 1024^2 render).  3 iterations: 1e-6.  100 iterations: an L1 loss under Adam flips sign(d) on
    1e-7 differences once texels converge, so two runs of the SAME loop on different thread counts already differ -- bulk agreement + equal loss."""
    from oracle import camera as ocam, optimize as oopt
    g = load_golden(name)
    cams = [ocam.Camera(p, int(g['cam_res'])) for p in g['cam_params']]
    uv_map, mask = oopt.texture_coordinates(cams, g['verts'], g['faces'], g['uvs'], g['mesh_tex_idx'], g['uv_centers'], g['uv_scales'], float(g['padding']),
                                            g['scale_factors'], 1024)
    shr = g['shrinked'] if g['shrinked'].size else None
    atlas, images = oopt.optimize_color(g['atlas0'], g['inpainted'], uv_map, mask, shr, iterations=int(g['iterations']))
    d = np.abs(atlas.numpy() - g['ref_atlas'])
    assert np.abs(g['ref_atlas'][0] - g['atlas0']).max() > 0.1                 # the reference loop moved the atlas
    if tol is not None:
        assert d.max() <= tol, d.max()
        assert np.abs(images[:, :, ::16, ::16].numpy() - g['ref_images_small']).max() <= 1e-6
    else:
        assert (d <= 1e-3).mean() > 0.97, (d <= 1e-3).mean()
        assert np.abs(images.mean(dim=(2, 3)).numpy() - g['ref_images_mean']).max() <= 2e-3
    untouched = (g['ref_atlas'][0] == g['atlas0']).all(0)
    assert untouched.any() and np.array_equal(atlas.numpy()[0][:, untouched], g['atlas0'][:, untouched])
