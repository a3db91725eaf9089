import sys, time, torch
sys.path.insert(0, '/root/repo')
import pointdreamer_amd.ddnm_inpainting as di
dev = torch.device('cuda', 0)
for B in (int(sys.argv[1]),) if len(sys.argv) > 1 else (8, 16):
    inp = di.Inpainter(dev, ckpt_path=None, allow_random_weights=True, max_batch=B)
    inp.n_steps = 10
    x = torch.rand((B, 3, 256, 256), device=dev); m = (torch.rand((B, 256, 256), device=dev) > 0.7)
    for _ in range(2): inp.inpaint_views(x, m)
    torch.cuda.synchronize(); t = time.time(); inp.inpaint_views(x, m); torch.cuda.synchronize()
    dt = time.time() - t
    print(f"batch {B}: {dt*1e3:.1f} ms per 10 steps -> {dt*1e3/10/B*8:.2f} ms per forward-of-8-views")
    del inp; torch.cuda.empty_cache()
