"""End-to-end wall time of the shipped configs through the CLI entry point (demo.py:264-307, 455-466 in the reference: PLY in, per-view
PNGs + OBJ / MTL / atlas PNG out), VERDICT r4 item 8:
  * configs/nearest.yaml and configs/default.yaml (DDNM, random-init weights: no checkpoint offline),
  * one cloud (three runs: the first pays imports / LDS attribute calls / the stand-in geometry) and a directory of 8 clouds with
    --batch_shapes 4,
  * a split of one shape's wall time by wrapped call (GPU-synchronised around each call, so the split run itself is slower than the
    free-running ones): colorize_one_mesh, optimize_color, neighbour completion, the PNG / OBJ writers, the PLY reader.
Usage (GPU box): python tools/time_demo.py [--configs nearest default] [--ddnm-steps 100]  -> prints; tee it into profiles/r05_time_demo.txt"""
import argparse, collections, functools, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import demo, synthetic, io_utils, pipeline
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--configs', nargs='*', default=['nearest', 'default'])
ap.add_argument('--batch-shapes', type=int, default=4)
a = ap.parse_args()
tmp = tempfile.mkdtemp()
xyz, rgb = synthetic.sphere_points(30000, seed=3)
pc = os.path.join(tmp, 'ball.ply')
io_utils.save_colored_pc_ply(xyz, rgb, pc)
many = os.path.join(tmp, 'many'); os.makedirs(many)
for k in range(8):
    x2, c2 = synthetic.sphere_points(30000, seed=10 + k)
    io_utils.save_colored_pc_ply(x2, c2, os.path.join(many, f'ball{k}.ply'))


def run(cfg, target, out, extra=()):
    argv = ["--config", os.path.join(ROOT, "configs", cfg + ".yaml"), "--pc_file", target, "--set", f"output_path={out}"] + list(extra)
    if cfg != 'nearest':
        argv += ["--allow_random_weights"]
    torch.cuda.synchronize(); t = time.time()
    demo.main(argv)
    torch.cuda.synchronize()
    return time.time() - t


for cfg in a.configs:
    print(f"=== configs/{cfg}.yaml", flush=True)
    for i in range(3):
        print(f"one cloud, run {i}: demo.main wall {run(cfg, pc, f'{tmp}/{cfg}_one{i}') * 1e3:.1f} ms", flush=True)
    for i in range(2):
        w = run(cfg, many, f'{tmp}/{cfg}_dir{i}', ["--batch_shapes", str(a.batch_shapes)])
        print(f"directory of 8 clouds, --batch_shapes {a.batch_shapes}, run {i}: wall {w * 1e3:.1f} ms = {w / 8 * 1e3:.1f} ms per shape", flush=True)

# where one shape's wall time goes: synchronised wall time per wrapped call of one more single-cloud run
acc = collections.OrderedDict()


def wrap(mod, name):
    f = getattr(mod, name)

    @functools.wraps(f)
    def g(*args, **kw):
        torch.cuda.synchronize(); t = time.time(); r = f(*args, **kw); torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.time() - t
        return r
    setattr(mod, name, g)


for n in ('read_ply_xyzrgb', 'save_colored_pc_ply', 'savemeshtes2', 'save_CHW_RGB_img', 'save_CHW_RGBA_img', 'load_obj_mesh'):
    if hasattr(io_utils, n):
        wrap(io_utils, n)
for n in ('standin_geometry', 'save_textured_mesh', 'prepare'):
    if hasattr(demo, n):
        wrap(demo, n)
wrap(pipeline, 'colorize_one_mesh')
from pointdreamer_amd import optimize as popt, unproject as punp, ours_utils as pou
for mod, n in ((popt, 'optimize_color'), (punp, 'paint_invisible_areas_by_neighbors'), (punp, 'unproject'), (pou, 'get_sparse_images'),
               (pou, 'get_inpainted_images')):
    if hasattr(mod, n):
        wrap(mod, n)
io_utils.set_async(False) if hasattr(io_utils, 'set_async') else None
for cfg in a.configs:
    acc.clear()
    w = run(cfg, pc, f'{tmp}/{cfg}_split')
    print(f"=== split, configs/{cfg}.yaml, synchronous writers: wall {w * 1e3:.1f} ms; " + ", ".join(f"{k} {v * 1e3:.1f}" for k, v in acc.items()) + " (ms)", flush=True)
