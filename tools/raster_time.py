import sys, os; sys.path.insert(0, '/root/repo')
from pointdreamer_amd import _lib as _l
if len(sys.argv) > 1: _l.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch, time
from pointdreamer_amd import synthetic, ours_utils as ou
import pointdreamer_amd.camera_utils as cu
dev='cuda:0'
sh = synthetic.make_shape(30000, 1024)
cams, base_dirs, eyes2, ups = cu.create_cameras(8, 1.6, 512, device=dev)
Tn = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
V, F, P = Tn(sh['vertices']), Tn(sh['faces']), Tn(sh['points'])
for _ in range(3): r = ou.get_rendered_hard_mask_and_face_idx_batch(cams, V, F, P, None, True, 0.05)
torch.cuda.synchronize(); t=time.time()
for _ in range(20): r = ou.get_rendered_hard_mask_and_face_idx_batch(cams, V, F, P, None, True, 0.05)
torch.cuda.synchronize(); print('P1+P2 us', (time.time()-t)/20*1e6)
if len(sys.argv) > 2:                                   # save the outputs for a comparison between builds
    np.savez(sys.argv[2], **{f'o{i}': (x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)) for i, x in enumerate(r) if torch.is_tensor(x)})
