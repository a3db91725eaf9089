"""Host-side cost of one demo shape (PLY in, OBJ/MTL/PNG tree out) around the GPU work: runs the CLI entry point on a synthetic
cloud three times with configs/nearest.yaml (so the diffusion is out of the way) and prints the reference's own timing log lines."""
import os, sys, tempfile, time, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointdreamer_amd import demo, synthetic, io_utils
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
xyz, rgb = synthetic.sphere_points(30000, seed=3)
pc = os.path.join(tmp, 'ball.ply')
io_utils.save_colored_pc_ply(xyz, rgb, pc)
many = os.path.join(tmp, 'many'); os.makedirs(many)
for k in range(8):
    x2, c2 = synthetic.sphere_points(30000, seed=10 + k)
    io_utils.save_colored_pc_ply(x2, c2, os.path.join(many, f'ball{k}.ply'))
extra = sys.argv[1:]
for i in range(3):
    t = time.time()
    demo.main(["--config", os.path.join(ROOT, "configs", "nearest.yaml"), "--pc_file", pc, "--set", f"output_path={tmp}/out{i}"] + extra)
    torch.cuda.synchronize()
    print(f"run {i}: demo.main wall {time.time() - t:.3f} s")
t = time.time()
demo.main(["--config", os.path.join(ROOT, "configs", "nearest.yaml"), "--pc_file", many, "--set", f"output_path={tmp}/outm"] + extra)
print(f"directory of 8 clouds: demo.main wall {time.time() - t:.3f} s = {(time.time() - t) / 8 * 1e3:.1f} ms per shape")
# where the host time goes: wall time per wrapped call of the last run
import functools, collections
acc = collections.OrderedDict()
def wrap(mod, name):
    f = getattr(mod, name)
    @functools.wraps(f)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.time(); r = f(*a, **k); torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.time() - t
        return r
    setattr(mod, name, g)
from pointdreamer_amd import pipeline
for n in ('read_ply_xyzrgb', 'save_colored_pc_ply', 'savemeshtes2', 'save_CHW_RGB_img', 'save_CHW_RGBA_img', 'load_obj_mesh'):
    if hasattr(io_utils, n): wrap(io_utils, n)
for n in ('standin_geometry', 'save_textured_mesh', 'prepare'):
    wrap(demo, n)
wrap(pipeline, 'colorize_one_mesh')
t = time.time()
demo.main(["--config", os.path.join(ROOT, "configs", "nearest.yaml"), "--pc_file", pc, "--set", f"output_path={tmp}/outp"] + extra)
print(f"profiled run: wall {time.time() - t:.3f} s; " + ", ".join(f"{k} {v * 1e3:.1f} ms" for k, v in acc.items()))
