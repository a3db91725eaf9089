import sys, os
sys.path.insert(0, os.getcwd())
import torch
import pointdreamer_amd.ddnm_inpainting as di
from pointdreamer_amd import _lib
L = _lib.lib()
dev = torch.device('cuda:0')
sd = di.random_state_dict(dict(di.IMAGENET_256), seed=0)
N = int(sys.argv[1]); tile = int(sys.argv[2]); fuse = int(sys.argv[3])
m = di.UNetModel(max_batch=N, device=dev, **di.IMAGENET_256)
m.load_state_dict(sd)
x = torch.randn((N, 3, 256, 256), device=dev); t = torch.full((N,), 500.0, device=dev)
L.pdhip_debug_set_fuse_gn(fuse); L.pdhip_debug_set_conv_tile(tile)
out = m(x, t); torch.cuda.synchronize()
print('ok', N, tile, fuse, float(out.abs().mean()))
