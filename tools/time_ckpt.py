"""Time of Inpainter construction from a checkpoint on disk (the production start-up; the bench and the CLI tests use random weights):
a state dict with the reference's 566 key names / fp32 dtypes (2.2 GB, seeded random) is written to /tmp, then loaded twice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointdreamer_amd import ddnm_inpainting as di
sd = di.random_state_dict(dict(di.IMAGENET_256), seed=0)
sd = {k: v.float().cpu() for k, v in sd.items()}
path = '/tmp/256x256_diffusion_uncond.pt'
t = time.time(); torch.save(sd, path); print(f"wrote {os.path.getsize(path) / 1e9:.2f} GB in {time.time() - t:.2f} s", flush=True)
del sd
for i in range(2):
    torch.cuda.synchronize(); t = time.time()
    sdl = torch.load(path, map_location='cpu'); t1 = time.time()
    inp = di.Inpainter('cuda:0', state_dict=sdl, max_batch=8); torch.cuda.synchronize(); t2 = time.time()
    print(f"run {i}: torch.load {t1 - t:.2f} s, engine (arena + upload + f16 conversion + fused weights) {t2 - t1:.2f} s", flush=True)
    del inp, sdl
