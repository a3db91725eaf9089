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
# with the depth-test skip mask (the pipeline's call): only depth-rejected points are queried
from pointdreamer_amd import ours_utils as ou
sh = synthetic.make_shape(30000, 1024)
cams, base_dirs, eyes2, ups = cu.create_cameras(8, 1.6, 512, device=dev)
Tn = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
hard, fidx, depth, vuv, uvc, uvs, pad, puv, pdep = ou.get_rendered_hard_mask_and_face_idx_batch(cams, Tn(sh['vertices']), Tn(sh['faces']), Tn(sh['points']), None, True, 0.05)
vis0, _ = ou.get_point_validation_by_depth(512, puv, pdep, depth, offset=0.0001)
P = Tn(sh['points'])
for _ in range(2): got2 = hpr.hidden_point_removal(P, eyes2, 100, already_valid=vis0)
torch.cuda.synchronize(); t=time.time(); got2 = hpr.hidden_point_removal(P, eyes2, 100, already_valid=vis0); torch.cuda.synchronize(); print('skip-mask ms', (time.time()-t)*1e3)
full = hpr.hidden_point_removal(P, eyes2, 100)
print('OR identity', bool((got2 == (full | vis0)).all()))
