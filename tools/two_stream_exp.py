"""Experiment: does running two half-batches of the DDNM sampler on two HIP streams (two UNet handles, two host threads) beat one
full batch?  The MFMA-bound convs of one half could overlap the HBM-bound GroupNorm / 1x1 kernels of the other."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pointdreamer_amd.ddnm_inpainting as di
dev = torch.device('cuda', 0)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
V = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator().manual_seed(0)
imgs = torch.rand((V, 3, 256, 256), generator=g).to(dev); masks = (torch.rand((V, 256, 256), generator=g) > 0.8).float().to(dev)
sd = di.random_state_dict(dict(di.IMAGENET_256), seed=0)
one = di.Inpainter(dev, state_dict=sd, max_batch=V)
def t_one():
    torch.cuda.synchronize(); t = time.time(); one.inpaint_views(imgs, masks, n_steps=STEPS); torch.cuda.synchronize(); return time.time() - t
t_one(); a = min(t_one() for _ in range(2))
print(f"one stream, batch {V}: {a * 1e3 / STEPS:.2f} ms per DDNM step")
del one; torch.cuda.empty_cache()
halves = [di.Inpainter(dev, state_dict=sd, max_batch=V // 2) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
DELAY = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0       # ms of device-side delay on stream 1 (de-phases the two layer sequences)
def work(i):
    with torch.cuda.stream(streams[i]):
        if i == 1 and DELAY > 0: torch.cuda._sleep(int(DELAY * 2.0e6))
        halves[i].inpaint_views(imgs[i * V // 2:(i + 1) * V // 2], masks[i * V // 2:(i + 1) * V // 2], n_steps=STEPS)
def t_two():
    torch.cuda.synchronize(); t = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); return time.time() - t
t_two(); b = min(t_two() for _ in range(2))
print(f"delay {DELAY} ms: two streams, 2 x batch {V // 2}: {b * 1e3 / STEPS:.2f} ms per DDNM step ({(a / b - 1) * 100:+.1f} % throughput)")
