import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, time
from pointdreamer_amd import synthetic, hpr
import pointdreamer_amd.camera_utils as cu
from oracle import project as oproj
dev='cuda:0'
pts,_ = synthetic.sphere_points(30000, seed=0)
_,_,eyes,_ = cu.create_cameras(8,1.6,512,device=dev)
T=torch.from_numpy(pts).to(dev)
for _ in range(2): got = hpr.hidden_point_removal(T, eyes, 100)
torch.cuda.synchronize(); t=time.time(); got = hpr.hidden_point_removal(T, eyes, 100); torch.cuda.synchronize(); print('full ms', (time.time()-t)*1e3)
want = oproj.point_validation_by_hpr(pts, eyes, 100)
g = got.cpu().numpy()
print('mismatch', (g!=want).mean(), 'false-visible', (g & ~want).mean(), 'false-hidden', (~g & want).mean(), 'visible frac', want.mean())
