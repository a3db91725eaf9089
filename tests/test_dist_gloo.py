"""Multi-process CPU tests (gloo, world_size 2 and 3) of the N>1 path: shard_range partitions, the single
all_gather that assembles the per-view inpainted images (including ragged view counts), and the view-parallel
colorize driver with injected CPU stage functions (the HIP stages need a GPU; the sharding logic does not)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def host_pack_bits(t):
    """The byte stream pdhip_pack_bits produces (bit b of byte j = element 8 j + b), in torch on the host: injected with the host stages of the
    gloo tests (the product's own packer is the HIP kernel and refuses host tensors)."""
    flat = t.reshape(-1)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32)
    out = ((flat != 0).reshape(-1, 8).to(torch.int32) * w).sum(1).to(torch.uint8)
    return out.reshape(tuple(t.shape[:-1]) + (t.shape[-1] // 8,))


def host_unpack_bits(b, n):
    sh = torch.arange(8, dtype=torch.int32)
    out = ((b.reshape(-1).to(torch.int32).unsqueeze(1) >> sh) & 1).reshape(-1).to(torch.bool)
    return out.reshape(tuple(b.shape[:-1]) + (n,))


def _worker(rank, world, port, n_views, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pointdreamer_amd import dist as pdist
    try:
        mine = pdist.shard_range(n_views, rank, world)
        full = torch.arange(n_views * 3 * 4 * 4, dtype=torch.float32).reshape(n_views, 3, 4, 4)
        got = pdist.all_gather_views(full[mine.start:mine.stop].clone(), n_views, rank, world)
        assert torch.equal(got, full), "all_gather_views must reassemble the views in order on every rank"

        # the view-parallel driver with injected CPU stages: a rank owns project / inpaint / visibility of ITS views, one
        # all_gather of the per-view records, the cross-view stage replicated
        A, K, r = 8, 2, 4
        gen = torch.Generator().manual_seed(5)
        all_vis = torch.rand((n_views, A, A), generator=gen) > 0.5
        all_pk = torch.rand((K, n_views, A, A), generator=gen) > 0.5
        all_uvc = torch.rand((n_views, 1, 2), generator=gen)
        all_uvs = torch.rand((n_views, 1, 1), generator=gen) + 1
        all_sf = torch.rand((n_views,), generator=gen) + 0.5
        calls = {}

        def before(coords, colors, vertices, faces, cam_local, n_local, res, cam_res, save, opts, view_offset):
            assert n_local == len(mine) and view_offset == mine.start and cam_local['cams'] == list(mine)
            assert opts['point_validation_by_o3d'] is True          # same default as pipeline.colorize_one_mesh
            # demo.py:115-117: the optional depth-outlier refinement reaches the rank's own per-view stage (it used to be swallowed)
            assert opts['refine_point_validation'] is True and opts['refine_res'] == 256
            sl = slice(mine.start, mine.stop)
            return dict(sparse=full[sl].clone(), mask0=torch.ones_like(full[sl]), mask2=torch.ones_like(full[sl]),
                        scale_factors=all_sf[sl].clone(), uv_centers=all_uvc[sl].clone(), uv_scales=all_uvs[sl].clone(), padding=0.05,
                        mesh_depths=None)

        def inpaint(pre, save, inpainter, n_local, method, first_key, advance, view_offset):
            calls['inpaint'] = (n_local, first_key, advance)
            return pre['sparse'] * 2.0 + 1.0                          # stands in for the DDNM stage

        def visibility(pre, cam_local, cam_res, xatlas, kernels, save, view_offset):
            sl = slice(mine.start, mine.stop)
            return all_vis[sl].clone(), all_pk[:, sl].clone()

        def after(pre_all, inpainted, vis, pk, vertices, faces, f_normals, xatlas, camera_info, res, cam_res, kernels, cub, opt):
            assert torch.equal(vis, all_vis) and torch.equal(pk, all_pk)
            assert torch.equal(pre_all['uv_centers'], all_uvc) and torch.equal(pre_all['uv_scales'], all_uvs)
            assert torch.equal(pre_all['scale_factors'], all_sf) and pre_all['padding'] == 0.05
            return inpainted.sum(0).permute(1, 2, 0)

        atlas = pdist.colorize_one_mesh_view_parallel(
            None, None, None, None, None, dict(gb_pos=None, mask=torch.zeros((1, A, A, 1)), per_atlas_pixel_face_id=None),
            dict(cams=list(range(n_views)), base_dirs=None, eye_positions=None), n_views, r, 8, rank, world, shape_key=3,
            refine_point_validation_by_remove_abnormal_depth=True, refine_res=256,
            stages=dict(before=before, inpaint=inpaint, visibility=visibility, after=after, pack_bits=host_pack_bits, unpack_bits=host_unpack_bits))
        expect = (full * 2.0 + 1.0).sum(0).permute(1, 2, 0)
        assert torch.equal(atlas, expect), "every rank must hold the full atlas built from all views"
        assert calls['inpaint'] == (len(mine), 3 * n_views + mine.start, n_views), \
            "each rank inpaints only its own block of views, with noise keys = global view indices"
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_views", [(2, 8), (3, 8), (2, 5)])
def test_view_parallel_gloo(world, n_views):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_views, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == list(range(world))


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    from pointdreamer_amd import dist as pdist
    for n in (0, 1, 5, 8, 64):
        for w in (1, 2, 3, 4, 8):
            seen = []
            for r in range(w):
                seen += list(pdist.shard_range(n, r, w))
            assert seen == list(range(n))
            sizes = [len(pdist.shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_demo_directory_is_partitioned_over_ranks():
    """`demo.py --pc_file <dir>` under torch.distributed.run: every cloud goes to exactly one rank, blocks differ by at most one."""
    sys.path.insert(0, ROOT)
    from pointdreamer_amd import demo
    files = [f"c{i:02d}.ply" for i in range(11)]
    for w in (1, 2, 3, 8, 16):
        parts = [demo.files_of_rank(files, r, w) for r in range(w)]
        assert sum(parts, []) == files
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_view_parallel_rejects_more_ranks_than_views():
    """A rank without views would enter the per-view HIP stages with V = 0 while the others block in the all_gather: the driver
    refuses world > view_num up front (no process group needed: the check precedes every stage and the collective)."""
    sys.path.insert(0, ROOT)
    from pointdreamer_amd import dist as pdist
    xa = dict(gb_pos=torch.zeros((1, 4, 4, 3)), mask=torch.ones((1, 4, 4, 1), dtype=torch.bool), per_atlas_pixel_face_id=torch.zeros((1, 4, 4), dtype=torch.int64))
    with pytest.raises(ValueError, match="world size <= view_num"):
        pdist.colorize_one_mesh_view_parallel(None, None, None, None, None, xa, {}, view_num=2, res=4, cam_res=8, rank=2, world=3,
                                              texture_gen_method='nearest', stages={})


def test_all_gather_views_forced_collective_world1_gloo():
    """force_collective runs the all_gather at world size 1 too (the GPU suite does the same through RCCL)."""
    sys.path.insert(0, ROOT)
    from pointdreamer_amd import dist as pdist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        rec = torch.arange(24, dtype=torch.uint8).view(3, 8)
        out = pdist.all_gather_views(rec, 3, 0, 1, None, force_collective=True)
        assert torch.equal(out, rec) and out.data_ptr() != rec.data_ptr()
    finally:
        dist.destroy_process_group()


def test_view_record_is_bit_packed():
    """SURVEY 8(e) payload: image f32 + raw and K shrunk visibility maps at ONE BIT per texel + four crop parameters; pack / unpack are inverse
    and the byte stream is the NBF kernels' 64-texel words (bit b of byte j = texel 8 j + b)."""
    sys.path.insert(0, ROOT)
    from pointdreamer_amd import dist as pdist
    assert pdist.record_bytes((3, 256, 256), 1024, 1) == 786432 + 2 * 131072 + 16
    assert 8 * pdist.record_bytes((3, 256, 256), 1024, 1) < 9e6           # 8.4 MB per shape (round 5: 23 MB with byte maps)
    g = torch.Generator().manual_seed(0)
    V, A, K, r = 3, 16, 2, 4
    img = torch.randn((V, 3, r, r), generator=g)
    vis = torch.rand((V, A, A), generator=g) > 0.4
    pk = torch.rand((K, V, A, A), generator=g) > 0.6
    uvc, uvs, sf = torch.rand((V, 1, 2), generator=g), torch.rand((V, 1, 1), generator=g) + 1, torch.rand((V,), generator=g)
    rec = pdist.pack_view_records(img, vis, pk, uvc, uvs, sf, pack=host_pack_bits)
    assert rec.shape == (V, pdist.record_bytes((3, r, r), A, K)) and rec.dtype == torch.uint8
    i2, v2, p2, c2, s2, f2 = pdist.unpack_view_records(rec, (3, r, r), A, K, unpack=host_unpack_bits)
    assert torch.equal(i2, img) and torch.equal(v2, vis) and torch.equal(p2, pk) and torch.equal(c2, uvc) and torch.equal(s2, uvs) and torch.equal(f2, sf)
    b = host_pack_bits(torch.tensor([[1, 0, 0, 0, 0, 0, 0, 0, 0, 1] + [0] * 54], dtype=torch.uint8))
    with pytest.raises(Exception):
        pdist.pack_bits(torch.zeros((1, 64), dtype=torch.uint8))      # the product packer is the HIP kernel: host tensors are refused, not packed on the CPU
    assert b.shape == (1, 8) and b[0, 0] == 1 and b[0, 1] == 2 and int(b[0, 2:].sum()) == 0
