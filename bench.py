#!/usr/bin/env python3
"""Headline benchmark: shapes/hour of the project -> inpaint -> unproject texturing path
(BASELINE.json: 30k-point cloud, 8 x 256^2 views, DDNM 100 steps, 1024^2 atlas) on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one whole shape through the hot path with its inputs already resident in HBM:
P1-P6 project/sparse -> D1/U1 DDNM inpainting of all 8 views (100 UNet steps, views batched) -> N1-N3 NBF ->
Uq1-Uq5 unproject + dilate.  Shapes are independent, so ranks shard SHAPES with no data-path collective
(weak scaling: one shape per rank per step); `--parallel views` instead splits the 8 views of one shape across
ranks and assembles them with a single RCCL all_gather (strong scaling).
Random-init UNet weights of the reference architecture (the checkpoint cannot be fetched offline), synthetic shape.
Rank 0 prints ONE JSON line (contract in the task statement) incl. `roofline` (dominant kernel: the 3x3
implicit-GEMM conv, timed with HIP events on its launch stream inside the timed region) and `cpu_baseline`
(the oracle's CPU restatement of the reference's 'nearest' path on a bounded sample, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNET_FLOP_PER_FORWARD = 2.0 * 1119832768512            # 1 119 832 768 512 MAC per 256^2 forward (N=1)
PEAK_FP16_TFLOPS = 2500.0                              # MI355X dense f16/bf16 MFMA peak (MI355X_MICROARCH.md)


def _cpu_one_shape(sh, hpr=True):
    """One 8-view shape through the oracle's CPU restatement of the reference's 'nearest' path; returns per-stage seconds."""
    from oracle import camera as ocam, project as oproj, sparse as osparse, inpaint as oinp, unproject as ounp
    V = 8
    t = {}
    t0 = time.perf_counter()
    cams, base_dirs, eyes, ups = ocam.create_cameras(V, 1.6, 512)
    pr = oproj.project_batch(cams, sh['vertices'], sh['points'], True, 0.05)
    hard, fid, depth = oproj.rasterize(pr['pos'], sh['faces'], 512)
    hard_r = oproj.downsample_masks(hard, 256)
    vis, _ = oproj.point_validation_by_depth(512, pr['point_uvs'], pr['point_depths'], depth, 0.0001)
    t['project'] = time.perf_counter() - t0; t0 = time.perf_counter()
    if hpr:
        vis = vis | oproj.point_validation_by_hpr(sh['points'], eyes, 100)
    t['hpr'] = time.perf_counter() - t0; t0 = time.perf_counter()
    pp = oproj.point_pixels_for_res(pr['point_uvs'], 256)
    sp, m0, m2, sf = osparse.get_sparse_images(pp, sh['colors'], vis, hard_r, V, 256, 1, 1, 0.82)
    t['sparse'] = time.perf_counter() - t0; t0 = time.perf_counter()
    inp = np.stack([oinp.reference_nearest_inpaint_scipy(sp[i], m2[i]) for i in range(V)]).astype(np.float32)
    t['nearest_inpaint'] = time.perf_counter() - t0; t0 = time.perf_counter()
    o = ounp.unproject(inp, sh['f_normals'], 256, cams, 512, base_dirs, sh['gb_pos'], sh['mask'],
                       sh['per_atlas_pixel_face_id'], pr['uv_centers'], pr['uv_scales'], 0.05, sf, depth, [21], True)
    t['unproject_nbf'] = time.perf_counter() - t0; t0 = time.perf_counter()
    oinp.reference_nearest_inpaint_scipy(o['atlas_img'].transpose(2, 0, 1), sh['mask'][..., 0])
    t['dilate_atlas'] = time.perf_counter() - t0
    return t


def _cpu_worker(args):
    """One host core: 1 warm-up + `timed` shapes with hidden-point removal, then one without (the on / off split)."""
    seed, timed = args
    import torch as _t
    _t.set_num_threads(1)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:                                      # noqa: BLE001 -- optional: BLAS pools inside numpy / scipy
        pass
    from pointdreamer_amd import synthetic
    sh = synthetic.make_shape(30000, 1024, seed=seed)
    _cpu_one_shape(sh)
    t0 = time.perf_counter()
    stages = [_cpu_one_shape(sh) for _ in range(timed)]
    wall = time.perf_counter() - t0
    off = _cpu_one_shape(sh, hpr=False)
    return wall, stages, off


def cpu_baseline(timed=3, max_workers=None):
    """Reference CPU 'nearest' path next to the GPU number.  kind = "port": the reference has no runnable
    CPU pipeline (demo.py:12,19 hard-code CUDA; kaolin / nvdiffrast are CUDA-only), so this is the oracle's CPU restatement of
    it -- project, raster, depth test + hidden-point removal through qhull, sparse images, scipy griddata nearest inpaint, NBF
    unproject, atlas dilate -- at BASELINE sizes.  Host cores are used the way a CPU deployment of this throughput metric would
    use them: one independent shape stream per core (the per-shape code is serial numpy / scipy, as the reference's is), on ALL
    cores the process may use (scheduler affinity and cgroup CPU quota; capped by memory at ~1.5 GB per worker, or by `max_workers`); `cores`, `host_cores` and the
    per-core rate are reported.  Every worker runs 1 warm-up + `timed` shapes with hidden-point removal and one without (the bounded
    sample of the task statement: ~30 s of CPU work per core), the aggregate rate is shapes / slowest worker's wall time; the
    median seconds per shape over all workers x shapes is reported next to it."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    # cores this process may actually use: scheduler affinity and the cgroup CPU quota (the GPU boxes of the pool show 256 logical
    # CPUs but run the container under cpu.max = 16 CPUs: 256 workers there time-slice 16 cores and thrash -- 178 s per shape)
    usable = cores
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max', ):
        try:
            q, per = open(path).read().split()[:2]
            if q != 'max':
                usable = max(1, min(usable, int(int(q) / int(per) + 0.5)))
        except (OSError, ValueError):
            pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            usable = max(1, min(usable, int(q / per + 0.5)))
    except (OSError, ValueError):
        pass
    workers = usable if max_workers is None else max(1, min(usable, max_workers))
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                workers = max(1, min(workers, int(int(line.split()[1]) / (1.5 * 1024 * 1024))))
                break
    except OSError:
        pass
    model = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    ctx = mp.get_context('spawn')                          # the parent holds a live HIP runtime: never fork it
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_worker, [(100 + w, timed) for w in range(workers)])
    total_wall = time.perf_counter() - t0
    slowest = max(r[0] for r in res)
    value = workers * timed / slowest * 3600.0
    flat = [st for r in res for st in r[1]]
    med = {k: float(np.median([st[k] for st in flat])) for k in flat[0]}
    per_shape = float(np.median([sum(st.values()) for st in flat]))
    off = float(np.median([sum(r[2].values()) for r in res]))
    return dict(value=value, unit="shapes/hour", cores=workers, kind="port", cpu_model=model, host_cores=cores, usable_cores=usable,
                seconds_per_shape_one_core=per_shape, shapes_per_hour_one_core=3600.0 / per_shape,
                seconds_per_shape_hpr_off=off, value_hpr_off=value * per_shape / off,
                stage_seconds=med,
                sample=f"oracle CPU restatement of the reference's texture_gen_method='nearest' path (scipy griddata / qhull) on "
                       f"{workers} cores = every core the container may use (cgroup quota / affinity; the host shows {cores} logical CPUs; {model}): one independent 30k-point 8-view shape stream per core at A=1024, "
                       f"1 warm-up + {timed} timed shapes each = {workers * timed} shapes in {slowest:.1f} s (pool wall {total_wall:.1f} s); "
                       f"median {per_shape:.2f} s/shape/core with hidden-point removal, {off:.2f} s without; no diffusion on the CPU "
                       f"(a CPU fp32 UNet forward is ~8 s, x800 per DDNM shape).  Protocol note: SURVEY 8d asks for 3 warm-up + 10 timed shapes and "
                       f"the median; the task statement bounds this leg to a ~10-30 s sample, so every core runs 1 warm-up + {timed} timed shapes "
                       f"({workers * timed} timed shapes in all), value = aggregate rate over the slowest core, the per-shape figure is the median")


class LoadSampler:
    """Shader clock and board power sampled from sysfs WHILE the timed region runs (the evidence for the
    'power-limited ceiling' has to be read under load, not after it).  A daemon thread reads, at >= 5 Hz, the amdgpu hwmon / DPM files
    of the GPU this rank runs on (matched by PCI bus id; falls back to the first card): freq1_input (current sclk, Hz), the '*' line of
    pp_dpm_sclk, power1_average or power1_input (uW), once power1_cap.  Nothing here touches the GPU queue."""

    def __init__(self, dev, hz=10.0):
        import glob
        self.period = 1.0 / hz
        self.samples = []
        self.files = {}
        self._stop = False
        self._thread = None
        cards = sorted(glob.glob('/sys/class/drm/card[0-9]*/device'))
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = '%04x:%02x:%02x' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:                                   # noqa: BLE001 -- older torch: no PCI ids on the properties object
            pass
        pick = None
        for c in cards:
            real = os.path.realpath(c)
            if want and want in real.lower():
                pick = c
                break
        if pick is None:
            amd = [c for c in cards if os.path.exists(os.path.join(c, 'pp_dpm_sclk'))]
            pick = amd[0] if amd else None
        self.card = pick
        self.matched_pci = bool(want and pick and want in os.path.realpath(pick).lower())
        if pick:
            hw = sorted(glob.glob(os.path.join(pick, 'hwmon', 'hwmon*')))
            cand = {'freq': [os.path.join(h, 'freq1_input') for h in hw], 'dpm': [os.path.join(pick, 'pp_dpm_sclk')],
                    'power': [os.path.join(h, n) for h in hw for n in ('power1_average', 'power1_input')],
                    'cap': [os.path.join(h, 'power1_cap') for h in hw]}
            for k, fs in cand.items():
                for f in fs:
                    try:
                        open(f).read()
                        self.files[k] = f
                        break
                    except OSError:
                        continue

    def _read(self):
        row = {}
        try:
            if 'freq' in self.files:
                row['sclk_mhz'] = int(open(self.files['freq']).read()) / 1e6
            if 'dpm' in self.files:
                cur = [ln for ln in open(self.files['dpm']).read().splitlines() if ln.strip().endswith('*')]
                if cur:
                    row['dpm_mhz'] = float(''.join(ch for ch in cur[0].split(':')[1] if ch.isdigit() or ch == '.'))
            if 'power' in self.files:
                row['power_w'] = int(open(self.files['power']).read()) / 1e6
        except (OSError, ValueError):
            pass
        return row

    def _loop(self):
        while not self._stop:
            r = self._read()
            if r:
                self.samples.append(r)
            time.sleep(self.period)

    def start(self):
        import threading
        if self.files:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        out = dict(samples=len(self.samples), sample_hz=1.0 / self.period, card=self.card, matched_by_pci_bus_id=self.matched_pci,
                   source={k: os.path.basename(v) for k, v in self.files.items()})
        if self.samples and not any('sclk_mhz' in r for r in self.samples):     # no hwmon freq1_input: the DPM table's current level stands in
            for r in self.samples:
                if 'dpm_mhz' in r:
                    r['sclk_mhz'] = r['dpm_mhz']
            out['source']['freq'] = 'pp_dpm_sclk (current level)'
        for key, name in (('sclk_mhz', 'sclk_mhz_under_load'), ('dpm_mhz', 'dpm_sclk_mhz_under_load'), ('power_w', 'power_w_under_load')):
            v = [r[key] for r in self.samples if key in r]
            if v:
                out[name + '_mean'] = float(np.mean(v)); out[name + '_min'] = float(np.min(v)); out[name + '_max'] = float(np.max(v))
        try:
            if 'cap' in self.files:
                out['power_cap_w'] = int(open(self.files['cap']).read()) / 1e6
        except (OSError, ValueError):
            pass
        return out


def calibrate(dev, seconds=3.0):
    """In-run calibration of the box ("calibrate with a measured GEMM and a copy kernel, and report both"), untimed, after
    the timed region: the vendor's plain f16 GEMM (hipBLASLt through torch.matmul, 8192^3) on operands with the bench's statistics
    (activations ~ N(0, 1), weights ~ N(0, 0.05^2)) and on zeros -- the gap between the two is the data-dependent power limit --
    and a device-to-device copy of a 1 GiB f16 tensor (read + write).  sclk / power are read from sysfs when the box exposes them."""
    out = {}
    n = 8192
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def timed(fn, budget):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        it = max(5, min(400, int(budget * 1e3 / max(e0.elapsed_time(e1), 1e-3))))
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e-3
    try:
        a = torch.randn((n, n), device=dev).half(); b = (torch.randn((n, n), device=dev) * 0.05).half(); c = torch.empty((n, n), device=dev, dtype=torch.float16)
        t = timed(lambda: torch.matmul(a, b, out=c), seconds * 0.4)
        out['gemm_f16_random_tflops'] = 2.0 * n ** 3 / t / 1e12
        a.zero_(); b.zero_()
        t = timed(lambda: torch.matmul(a, b, out=c), seconds * 0.3)
        out['gemm_f16_zeros_tflops'] = 2.0 * n ** 3 / t / 1e12
        del a, b, c
        x = torch.empty((512 * 1024 * 1024,), device=dev, dtype=torch.float16).normal_(); y = torch.empty_like(x)
        t = timed(lambda: y.copy_(x), seconds * 0.2)
        out['copy_gbs'] = 2.0 * x.numel() * 2 / t / 1e9
        # the library's own 16-byte-per-lane copy kernel (pdhip_bench_copy16): the ceiling the GroupNorm passes are judged against
        import ctypes as C
        from pointdreamer_amd import _lib
        L = _lib.lib()
        best = {}
        for unroll in (1, 2, 4, 16 + 2, 16 + 4, 32 + 2, 32 + 4):     # (low bits: loads in flight; 16 +: nontemporal; 32 +: one slab per workgroup)
            for blocks in (16384, 65536, 262144):
                fn = lambda: L.pdhip_bench_copy16(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_longlong(x.numel() * 2), blocks, unroll,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
                t = timed(fn, seconds * 0.02)
                best[f"{('sweep', 'nt', 'slab')[unroll >> 4]}_u{unroll & 15}_b{blocks}"] = 2.0 * x.numel() * 2 / t / 1e9
        out['copy16_gbs'] = max(best.values())
        out['copy16_config'] = max(best, key=best.get)
        out['copy16_sweep_gbs'] = {k: round(v, 1) for k, v in best.items()}
        assert torch.equal(x[-4096:], y[-4096:])
        del x, y
    except Exception as e:                                  # noqa: BLE001 -- a side figure must not take the headline line down
        out['error'] = str(e)[:200]
    out['note'] = ("torch.matmul (hipBLASLt) 8192^3 f16 on N(0,1) x N(0,0.05^2) operands and on zeros; copy = 1 GiB f16 device-to-device through "
                   "torch's copy_ and through the library's own 16-byte-per-lane kernel (best of a small unroll / grid sweep), read + write "
                   "bytes; measured by this run after the timed region.  sclk / power under load are sampled DURING the timed region")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='ddnm', choices=['ddnm', 'nearest'])
    ap.add_argument('--parallel', default='shapes', choices=['shapes', 'views'])
    ap.add_argument('--ddnm-steps', type=int, default=100)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--profile-period', type=int, default=1, help='HIP events around the dominant kernel in every k-th UNet forward of the timed '
                    'region (1 = every forward, 0 = none: roofline.achieved is then null)')
    ap.add_argument('--no-graphs', action='store_true', help='nearest workload with several shapes per step: eager launches on streams '
                                                             'instead of one HIP graph per shape')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend (nccl = RCCL; gloo only for the one-GPU rehearsal of the N > 1 path)")
    ap.add_argument('--one-device', action='store_true', help='rehearsal of the N > 1 code path on a one-GPU box: every rank uses cuda:0 (with --backend gloo)')
    ap.add_argument('--no-extras', action='store_true', help='skip the nearest-workload / one-shape-latency side measurements')
    ap.add_argument('--extras-timeout', type=float, default=300.0, help='N > 1: seconds the view-parallel side figure may take before the headline line is printed without it')
    ap.add_argument('--shapes-per-step', type=int, default=4,
                    help='independent shapes textured per step on each GPU, their 8-view sets batched through the UNet together '
                         '(BASELINE configs[4] style); 1 = one shape at a time (configs[2], lowest latency).  What each setting measures on '
                         'one MI355X is in DESIGN.md section 8 (the default is the fastest)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from pointdreamer_amd import synthetic, pipeline, _lib, io_utils
    import pointdreamer_amd.camera_utils as cu
    # (torch sizes its intra-op CPU pool by the host's logical CPUs -- 256 on the pool's boxes -- while the container's cgroup grants 16:
    # an oversized OpenMP region exhausts the CFS quota and throttles the launch thread; the host side of a step is small tensors)
    torch.set_num_threads(max(1, min(8, io_utils.cpus_per_rank() // 2)))
    import pointdreamer_amd.ddnm_inpainting as di
    from pointdreamer_amd import dist as pdist
    _lib.lib()

    V, RES, CAM_RES, A = 8, 256, 512, 1024
    sh = synthetic.make_shape(30000, A, seed=rank if args.parallel == 'shapes' else 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    g = {k: T(v) for k, v in sh.items()}
    cams, base_dirs, eyes, ups = cu.create_cameras(V, 1.6, CAM_RES, device=dev)
    camera_info = dict(cams=cams, base_dirs=base_dirs, eye_positions=eyes, up_dirs=ups)
    xatlas = dict(gb_pos=g['gb_pos'], mask=g['mask'], per_atlas_pixel_face_id=g['per_atlas_pixel_face_id'], uvs=None, mesh_tex_idx=None)
    if world > 1 and args.parallel == 'shapes':            # the view-parallel extra textures the SAME shape on every rank (seed 0)
        sh0 = synthetic.make_shape(30000, A, seed=0)
        g0 = {k: T(v) for k, v in sh0.items()}
        xatlas0 = dict(gb_pos=g0['gb_pos'], mask=g0['mask'], per_atlas_pixel_face_id=g0['per_atlas_pixel_face_id'], uvs=None, mesh_tex_idx=None)
    else:
        g0, xatlas0 = g, xatlas
    inpainter = None
    SPS = max(1, args.shapes_per_step) if args.parallel == 'shapes' else 1
    views_here = V * SPS if args.parallel == 'shapes' else len(pdist.shard_range(V, rank, world))
    if args.workload == 'ddnm':
        inpainter = di.Inpainter(dev, ckpt_path=None, allow_random_weights=True, max_batch=views_here)
        inpainter.n_steps = args.ddnm_steps
    method = 'DDNM_inpaint' if args.workload == 'ddnm' else 'nearest'
    cfg = dict(view_num=V, res=RES, cam_res=CAM_RES, point_validation_by_o3d=True, texture_gen_method=method, point_size=1,
               edge_point_size=1, crop_img=True, crop_padding=0.05, mask_ratio_thresh=0.82, optimize_from=None,
               edge_dilate_kernels=[21], complete_unseen_by='unproject', inpainter=inpainter)

    extra = []
    for k in range(1, SPS):                      # further independent shapes of this rank (same mesh / atlas, different clouds)
        sk = synthetic.make_shape(30000, A, seed=1000 * k + rank)
        extra.append(dict(coords=T(sk['points']), colors=T(sk['colors'])))
    batch = [dict(coords=g['points'], colors=g['colors'], vertices=g['vertices'], faces=g['faces'], f_normals=g['f_normals'], xatlas=xatlas)]
    batch += [dict(coords=e['coords'], colors=e['colors'], vertices=g['vertices'], faces=g['faces'], f_normals=g['f_normals'], xatlas=xatlas)
              for e in extra]

    sgraphs = None
    if args.workload == 'nearest' and SPS > 1 and args.parallel == 'shapes' and not args.no_graphs:
        # the geometry-only workload is launch-bound (~45 launches of 4-100 us per shape): one HIP graph per shape slot, SPS slots on
        # streams of their own (pipeline.ShapeGraphs, bit-identical to the eager path)
        gcfg = {k: v for k, v in cfg.items() if k not in ('view_num', 'res', 'cam_res', 'inpainter')}
        sgraphs = pipeline.ShapeGraphs(SPS, 30000, g['vertices'], g['faces'], g['f_normals'], xatlas, camera_info, V, RES, CAM_RES, **gcfg)
        gclouds = [(b_['coords'], b_['colors']) for b_ in batch]

    def step():
        if sgraphs is not None:
            return sgraphs.run(gclouds, clone=False)
        if SPS > 1:
            return pipeline.colorize_meshes_batched(batch, camera_info, **{k: v for k, v in cfg.items()
                                                                          if k not in ('optimize_from', 'complete_unseen_by')})
        if args.parallel == 'views' and world > 1:
            return pdist.colorize_one_mesh_view_parallel(g['points'], g['colors'], g['vertices'], g['faces'], g['f_normals'],
                                                         xatlas, camera_info, rank=rank, world=world, **cfg)
        return pipeline.colorize_one_mesh(g['points'], g['colors'], g['vertices'], g['faces'], g['f_normals'], xatlas,
                                          camera_info, **cfg)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if inpainter is not None:
        inpainter.model.profile(args.profile_period)
    sampler = LoadSampler(dev) if rank == 0 else None
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    under_load = sampler.stop() if sampler is not None else None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    shapes = args.steps * (world * SPS if args.parallel == 'shapes' else 1)
    value = shapes / dt * 3600.0

    roofline = None
    extras = {}

    print_lock = threading.Lock()
    printed = [False]

    def finish(extras_, from_watchdog=False):
        # ONE headline line per run, whoever gets here first (a side figure finishing right at its timeout could otherwise
        # race the watchdog's copy of the line); the guard is taken before any slow work so that the loser does nothing
        with print_lock:
            if printed[0]:
                return
            printed[0] = True
        calib = None
        if rank == 0 and inpainter is not None and not args.no_extras and not from_watchdog:
            calib = calibrate(dev)
            if roofline is not None and calib.get('gemm_f16_random_tflops'):
                roofline['calibrated_peak'] = calib['gemm_f16_random_tflops']
                roofline['frac_of_calibrated'] = roofline['achieved'] / calib['gemm_f16_random_tflops'] if roofline.get('achieved') else None
                roofline['calibration'] = calib
        if roofline is not None and under_load is not None:
            roofline.setdefault('calibration', {})['under_load'] = under_load
            for k in ('sclk_mhz_under_load_mean', 'sclk_mhz_under_load_min', 'power_w_under_load_mean', 'power_cap_w'):
                if k in under_load:
                    roofline['calibration'][k if 'sclk' in k else k.replace('_under_load', '')] = under_load[k]
            clk = under_load.get('sclk_mhz_under_load_mean') or under_load.get('dpm_sclk_mhz_under_load_mean')
            if clk and roofline.get('achieved'):
                # the dense peak is quoted at the 2 400 MHz maximum clock: the same matrix pipes at the clock the timed region actually ran at
                roofline['clock_adjusted_peak'] = PEAK_FP16_TFLOPS * clk / 2400.0
                roofline['frac_of_clock_adjusted_peak'] = roofline['achieved'] / roofline['clock_adjusted_peak']
        if world > 1 and not from_watchdog:
            dist.barrier()
        if rank == 0:
            out = dict(metric=f"shapes/hour (30k-pt cloud, 8x256^2 views, {'DDNM' if args.workload == 'ddnm' else 'nearest inpainting, no diffusion'}) on MI355X", value=value, unit="shapes/hour",
                       n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
                       higher_is_better=True, scaling="weak" if args.parallel == 'shapes' else "strong", vs_baseline=None,
                       dtype="f16 (f32 accumulate; f32 GroupNorm/softmax statistics, f32 geometry)", data="synthetic",
                       config=dict(workload=f"configs[2]: synthetic 30k-point sphere shape, 8x256^2 views, texture_gen_method="
                                            f"'{method}' ({args.ddnm_steps} DDNM steps, 552.8M-param guided-diffusion UNet, random-init weights), "
                                            f"NBF [21], atlas 1024^2, complete_unseen_by='unproject', optimize_from=None, hidden-point removal on (device)" + (f"; {SPS} independent shapes per step, their views batched through the UNet together" if SPS > 1 else ""),
                                   parallelism=f"{args.parallel}-parallel x{world}", views_per_unet_batch=views_here,
                                   shapes_per_step=world * SPS if args.parallel == 'shapes' else 1),
                       roofline=roofline, extras=extras_ or None)
            if world == 1 and not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline()
            else:
                out['cpu_baseline'] = None
            print(json.dumps(out), flush=True)

    if inpainter is not None:
        ms, flops, launches = inpainter.model.profile_read()
        ams, aflops, alaunches = inpainter.model.profile_read(attention=True)
        inpainter.model.profile(False)
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic, traffic_source = None, None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_conv.json')))
        pmc = cands[-1] if cands else ''
        if pmc:          # HBM bytes per launch from separate rocprofv3 --pmc passes of this same command (tools/pmc_bench.sh)
            pj = json.load(open(pmc))                 # per-launch bytes depend on the UNet batch: only quoted for the batch it was measured at
            traffic = pj.get('hbm_bytes_per_launch') if pj.get('shapes_per_step', 1) == SPS else None
            if traffic is not None:                   # NOT measured by this run: PMC counters need their own rocprofv3 passes
                traffic_source = (f"{os.path.relpath(pmc, ROOT)} (rocprofv3 --pmc passes of this command, recorded earlier; FETCH_SIZE "
                                  f"doubled per the gfx950 correction + WRITE_SIZE)")
        roofline = dict(bound="mfma", kernel="k_conv3x3_halo (3x3 conv of the 64^2/128^2/256^2 levels, LDS-resident activation halo, f16 in / f32 acc)", achieved=achieved,
                        peak=PEAK_FP16_TFLOPS, unit="TFLOP/s", frac=achieved / PEAK_FP16_TFLOPS, traffic=traffic, traffic_source=traffic_source,
                        whole_unet_frac=(UNET_FLOP_PER_FORWARD * views_here * args.ddnm_steps * args.steps) / dt / 1e12 / PEAK_FP16_TFLOPS,
                        launches=int(launches), avg_launch_ms=ms / max(launches, 1),
                        flops_per_launch=flops / max(launches, 1),
                        unet_forward_tflops_effective=(UNET_FLOP_PER_FORWARD * views_here * args.ddnm_steps * args.steps) / dt / 1e12,
                        attention=dict(bound="mfma", kernel="k_attention_t64 / k_attention (QK^T, softmax, PV; 32^2, 16^2, 8^2 levels)",
                                       achieved=(aflops / (ams * 1e-3) / 1e12) if ams > 0 else None, peak=PEAK_FP16_TFLOPS,
                                       unit="TFLOP/s", frac=(aflops / (ams * 1e-3) / 1e12 / PEAK_FP16_TFLOPS) if ams > 0 else None,
                                       launches=int(alaunches), avg_launch_ms=ams / max(alaunches, 1),
                                       flops_per_launch=aflops / max(alaunches, 1)))
        # driver-timed side figures on the same box, after the timed region (world 1 only): BASELINE configs[1] ('nearest' + NBF,
        # one shape per step) and the latency of ONE shape through the DDNM path (configs[2] at one shape per step)
        if world == 1 and not args.no_extras:
            one = lambda c: pipeline.colorize_one_mesh(g['points'], g['colors'], g['vertices'], g['faces'], g['f_normals'], xatlas,
                                                       camera_info, **c)
            cn = dict(cfg, texture_gen_method='nearest', inpainter=None)
            for _ in range(3):
                one(cn)
            sync(); t1 = time.perf_counter()
            for _ in range(50):
                one(cn)
            sync(); dn = (time.perf_counter() - t1) / 50
            cno = dict(cn, point_validation_by_o3d=False)
            for _ in range(3):
                one(cno)
            sync(); t1 = time.perf_counter()
            for _ in range(50):
                one(cno)
            sync(); dno = (time.perf_counter() - t1) / 50
            extras['nearest'] = dict(metric="shapes/hour (configs[1]: 30k-pt cloud, 8x256^2 views, texture_gen_method='nearest' + NBF [21], "
                                            "hidden-point removal on, one shape per step)", value=3600.0 / dn, ms_per_shape=dn * 1e3,
                                     ms_per_shape_hpr_off=dno * 1e3, value_hpr_off=3600.0 / dno,
                                     # rows P1-Uq5 without P3b move 254 MB of algorithmic bytes per shape
                                     roofline=dict(bound="hbm", achieved=254e6 / dno / 1e9, peak=8000.0, unit="GB/s",
                                                   frac=254e6 / dno / 1e9 / 8000.0, traffic=None,
                                                   note="aggregate over the ~21 launches of one shape, hidden-point removal off"))
            # configs[4]-style geometry: 8 shapes per step, each shape's ~45 launches replayed as ONE HIP graph on a stream of its own
            # (pipeline.ShapeGraphs; results bit-identical to the eager path)
            try:
                gcfg = {k: v for k, v in cn.items() if k not in ('view_num', 'res', 'cam_res', 'inpainter')}
                sg = pipeline.ShapeGraphs(8, 30000, g['vertices'], g['faces'], g['f_normals'], xatlas, camera_info, V, RES, CAM_RES, **gcfg)
                cl = []
                for k in range(8):
                    sk = synthetic.make_shape(30000, A, seed=7000 + k)
                    cl.append((T(sk['points']), T(sk['colors'])))
                for _ in range(3):
                    sg.run(cl, clone=False)
                sync(); t1 = time.perf_counter()
                for _ in range(30):
                    sg.run(cl, clone=False)
                sync(); dg = (time.perf_counter() - t1) / 30 / 8
                extras['nearest_graphs'] = dict(metric="shapes/hour (configs[1] workload, 8 independent shapes per step, one HIP graph per shape on 8 "
                                                       "streams, hidden-point removal on)", value=3600.0 / dg, ms_per_shape=dg * 1e3)
                del sg
            except Exception as e:                      # noqa: BLE001 -- a side figure must not take the headline line down
                extras['nearest_graphs'] = dict(error=str(e)[:200])
            # the same 8 shapes with ONE launch per stage for all of them (round 4: pdhip_*_shapes, pointdreamer_amd/shapes.py): per-shape
            # inputs stacked [S, ...] -- the stacking of the 8 shape dicts (clouds, mesh and atlas maps: one copy each) is inside the timed
            # region, as colorize_meshes_batched does it for a directory run; `geometry_kept` re-stacks only the clouds
            try:
                from pointdreamer_amd import shapes as shp
                b8 = [dict(coords=c_[0], colors=c_[1], vertices=g['vertices'], faces=g['faces'], f_normals=g['f_normals'], xatlas=xatlas) for c_ in cl]
                skw = dict(texture_gen_method='nearest', point_size=1, edge_point_size=1, crop_img=True, crop_padding=0.05, mask_ratio_thresh=0.82,
                           edge_dilate_kernels=[21], point_validation_by_o3d=True)
                run8 = lambda: shp.colorize_shapes(shp.stack(b8), camera_info, V, RES, CAM_RES, **skw)
                for _ in range(3):
                    run8()
                sync(); t1 = time.perf_counter()
                for _ in range(30):
                    run8()
                sync(); ds8 = (time.perf_counter() - t1) / 30 / 8
                st8 = shp.stack(b8)
                def run8k():
                    st8['coords'] = torch.stack([c_[0] for c_ in cl], 0); st8['colors'] = torch.stack([c_[1] for c_ in cl], 0)
                    return shp.colorize_shapes(st8, camera_info, V, RES, CAM_RES, **skw)
                for _ in range(3):
                    run8k()
                sync(); t1 = time.perf_counter()
                for _ in range(30):
                    run8k()
                sync(); dk8 = (time.perf_counter() - t1) / 30 / 8
                extras['nearest_stacked'] = dict(metric="shapes/hour (configs[1] workload, 8 independent shapes per step, ONE launch per stage for all 64 "
                                                        "views (pdhip_*_shapes), hidden-point removal on)", value=3600.0 / ds8, ms_per_shape=ds8 * 1e3,
                                                 ms_per_shape_geometry_kept=dk8 * 1e3)
                del st8
                # the same with RAGGED meshes (configs[4] with real data: every shape its own mesh -- 8 820 ... 11 220 faces -- and chart mask;
                # shapes.stack pads vertices / faces to the largest, the stages stay one launch each)
                rg = []
                for k, (st_, sl_) in enumerate(((45, 98), (50, 100), (47, 104), (52, 96), (49, 110), (55, 102), (46, 100), (51, 108))):
                    sk = synthetic.make_shape(30000, A, stacks=st_, slices=sl_, seed=7100 + k, gutter=2 + (k % 3))
                    rg.append(dict(coords=T(sk['points']), colors=T(sk['colors']), vertices=T(sk['vertices']), faces=T(sk['faces']), f_normals=T(sk['f_normals']),
                                   xatlas=dict(gb_pos=T(sk['gb_pos']), mask=T(sk['mask']), per_atlas_pixel_face_id=T(sk['per_atlas_pixel_face_id']))))
                assert shp.uniform(rg) and shp.ragged(rg)
                runr = lambda: shp.colorize_shapes(shp.stack(rg), camera_info, V, RES, CAM_RES, **skw)
                for _ in range(3):
                    runr()
                sync(); t1 = time.perf_counter()
                for _ in range(30):
                    runr()
                sync(); dr8 = (time.perf_counter() - t1) / 30 / 8
                extras['nearest_stacked']['ms_per_shape_ragged_meshes'] = dr8 * 1e3
                extras['nearest_stacked']['ragged_faces'] = [int(x_['faces'].shape[0]) for x_ in rg]
                del rg
            except Exception as e:                      # noqa: BLE001
                extras['nearest_stacked'] = dict(error=str(e)[:200])
            one(cfg)
            d1s = []
            for _ in range(3):
                sync(); t1 = time.perf_counter()
                one(cfg)
                sync(); d1s.append(time.perf_counter() - t1)
            d1 = float(np.median(d1s))
            extras['ddnm_one_shape'] = dict(metric="seconds per shape, configs[2] with one shape per step (UNet batch 8); median of 3", seconds=d1,
                                            samples=d1s, value=3600.0 / d1)
            # view-parallel PROJECTION from this GPU's own latencies (configs[3] cannot be measured on one GPU): a rank of an 8-GPU run
            # inpaints ONE view (UNet batch 1), of a 4- / 2-GPU run two / four -- 20 sampler steps each, timed here; everything that is
            # not the sampler (per-view geometry, the all-gather, the replicated blend) is taken from the one-shape run above
            try:
                ng = args.ddnm_steps
                vp = {}
                g_ = torch.Generator(device='cpu').manual_seed(3)
                for gpus, b in ((8, 1), (4, 2), (2, 4), (1, 8)):
                    im = torch.rand((b, 3, RES, RES), generator=g_).to(dev); mk = (torch.rand((b, RES, RES), generator=g_) > 0.5).float().to(dev)
                    inpainter.inpaint_views(im * mk[:, None], mk, n_steps=3)
                    sync(); t1 = time.perf_counter()
                    inpainter.inpaint_views(im * mk[:, None], mk, n_steps=20)
                    sync(); vp[gpus] = (time.perf_counter() - t1) / 20
                rest = max(d1 - ng * vp[1], 0.0)                       # one-shape time that is not the batch-8 sampler
                extras['view_parallel_projection'] = dict(
                    metric="configs[3] projected from single-GPU latencies: one-shape seconds / (ddnm_steps x sampler step at V / G views per rank + the "
                           "non-sampler rest of a shape); NOT a measurement of N > 1 hardware",
                    sampler_step_ms={f"batch_{8 // k_}": v_ * 1e3 for k_, v_ in vp.items()}, non_sampler_seconds=rest,
                    projected_speedup={f"{k_}_gpus": d1 / (ng * v_ + rest) for k_, v_ in vp.items() if k_ > 1})
            except Exception as e:                      # noqa: BLE001
                extras['view_parallel_projection'] = dict(error=str(e)[:200])
        # the north_star's scaling claim (configs[3]): ONE shape, its 8 views sharded over the ranks, one RCCL all_gather -- run after the
        # timed region on every rank, next to the same shape on rank 0 alone (UNet batch 8), so the line carries the speed-up itself
        if world > 1 and args.parallel == 'shapes' and not args.no_extras and world <= V:
            # (this side figure is the only collective after the timed region and has never met N > 1 hardware: if it does not come
            # back, every rank leaves through the watchdog and rank 0 still prints the headline line)
            def abandon():
                #  the exit must not depend on finish() returning: an exception in the timer thread would leave rank 0 in the
                # collective after the other ranks have gone.  Every rank leaves with code 0 -- the headline was measured and printed, and a
                # non-zero rank would make the launcher report the whole run as failed -- but says so on stderr, so a log shows that the
                # side figure's collective really did not come back
                try:
                    print(f"[bench rank {rank}] view-parallel side figure abandoned after {args.extras_timeout:.0f} s", file=sys.stderr, flush=True)
                    if rank == 0:
                        finish(dict(extras, view_parallel=dict(
                            error=f"no result within {args.extras_timeout:.0f} s: side figure abandoned, headline unaffected")), True)
                finally:
                    sys.stdout.flush()
                    os._exit(0)
            watchdog = threading.Timer(args.extras_timeout, abandon)
            watchdog.daemon = True
            watchdog.start()
            try:
                vp = lambda: pdist.colorize_one_mesh_view_parallel(g0['points'], g0['colors'], g0['vertices'], g0['faces'], g0['f_normals'],
                                                                  xatlas0, camera_info, rank=rank, world=world, **cfg)
                vp(); vps = []
                for _ in range(3):
                    sync(); t1 = time.perf_counter()
                    vp()
                    sync(); tt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    vps.append(float(tt.item()))
                dvp = float(np.median(vps))
                d1 = None
                if rank == 0:
                    one = lambda c: pipeline.colorize_one_mesh(g0['points'], g0['colors'], g0['vertices'], g0['faces'], g0['f_normals'], xatlas0,
                                                               camera_info, **c)
                    one(cfg); d1s = []
                    for _ in range(3):
                        torch.cuda.synchronize(); t1 = time.perf_counter()
                        one(cfg)
                        torch.cuda.synchronize(); d1s.append(time.perf_counter() - t1)
                    d1 = float(np.median(d1s))
                sync()
                extras['view_parallel'] = dict(metric=f"configs[3]: one shape, 8 views sharded over {world} GPUs ({len(pdist.shard_range(V, 0, world))} per rank), "
                                                      "one RCCL all_gather of the per-view records; median of 3, max over ranks",
                                               seconds_per_shape=dvp, samples=vps, n1_one_shape_seconds=d1,
                                               speedup_vs_n1_one_shape=(d1 / dvp) if d1 else None)
            except Exception as e:                          # noqa: BLE001 -- a side figure must not take the headline line down
                extras['view_parallel'] = dict(error=str(e)[:300])
            watchdog.cancel()
    else:
        roofline = dict(bound="hbm", kernel="n/a (nearest workload: sub-millisecond HBM-bound kernels)", achieved=None, peak=8000.0,
                        unit="GB/s", frac=None, traffic=None)
    finish(extras)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
