"""cProfile of a directory run of the CLI (configs/nearest.yaml, 8 clouds, --batch_shapes 4): where the HOST time of the batched route goes."""
import cProfile, pstats, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import demo, synthetic, io_utils
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
many = os.path.join(tmp, 'many'); os.makedirs(many)
for k in range(8):
    x2, c2 = synthetic.sphere_points(30000, seed=10 + k)
    io_utils.save_colored_pc_ply(x2, c2, os.path.join(many, f'ball{k}.ply'))
argv = lambda o: ["--config", os.path.join(ROOT, "configs", "nearest.yaml"), "--pc_file", many, "--set", f"output_path={tmp}/{o}", "--batch_shapes", "4"]
demo.main(argv('w0')); demo.main(argv('w1'))
torch.cuda.synchronize()
pr = cProfile.Profile()
t = time.time(); pr.enable(); demo.main(argv('p')); torch.cuda.synchronize(); pr.disable()
print('wall', time.time() - t)
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
