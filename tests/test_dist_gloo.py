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

        calls = {'inpaint_views': None}

        def project(coords, colors, *a):
            return dict(sparse=full.clone(), mask0=torch.ones_like(full), mask2=torch.ones_like(full),
                        scale_factors=torch.ones(n_views), uv_centers=None, uv_scales=None, padding=0.05, mesh_depths=None)

        def inpaint(sparse, m0, m2, save_path, inpainter, view_num, method):
            calls['inpaint_views'] = view_num
            return sparse * 2.0 + 1.0                                  # stands in for the DDNM stage

        def unproject(inpainted, *a):
            return inpainted.sum(0).permute(1, 2, 0), None, None, None, None

        atlas = pdist.colorize_one_mesh_view_parallel(
            None, None, None, None, None, dict(gb_pos=None, mask=None, per_atlas_pixel_face_id=None),
            dict(cams=None, base_dirs=None, eye_positions=None), n_views, 4, 8, rank, world,
            stages=dict(project=project, inpaint=inpaint, unproject=unproject, dilate=lambda a, m: a))
        expect = (full * 2.0 + 1.0).sum(0).permute(1, 2, 0)
        assert torch.equal(atlas, expect), "every rank must hold the full atlas built from all views"
        assert calls['inpaint_views'] == len(mine), "each rank inpaints only its own block of views"
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
