import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointdreamer_amd import _lib as _l
if len(sys.argv) > 1: _l.LIB_PATH = os.path.abspath(sys.argv[1])          # alternative (lab) build
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
# lab build with -DPD_HPR_STATS: round statistics of the last call
import ctypes as C
from pointdreamer_amd import _lib
try:
    fn = C.CDLL(_lib.LIB_PATH).pdhip_lab_hpr_stats
    buf = (C.c_ulonglong * 48)()
    fn(buf, 1); got2 = hpr.hidden_point_removal(P, eyes2, 100, already_valid=vis0); torch.cuda.synchronize(); fn(buf, 1)
    r = max(buf[44], 1)
    print(f'level 1, cycles per wave round (s_memtime): B operands {buf[40] / r:.0f}, scan {buf[41] / r:.0f}, re-evaluation + join {buf[42] / r:.0f}, gather + step {buf[43] / r:.0f}; wave rounds {buf[44]}')
    fr = max(buf[19], 1)
    print(f'level 2, cycles per query round (rounds that reach the solve): working-set scan {buf[45] / fr:.0f}, weak test + global scan {buf[46] / fr:.0f}, verdict + solve {buf[47] / fr:.0f}')
    nm = max(buf[18] - buf[35], 1)
    print(f'level 2 per-query wall time (100 MHz ticks -> us): mean {buf[32] / max(buf[18], 1) / 100:.1f}, max {buf[33] / 100:.1f}; members {buf[35]}: mean {buf[34] / max(buf[35], 1) / 100:.1f}; others {nm}: mean {(buf[32] - buf[34]) / nm / 100:.1f}')
    print('level 2: max scans per query', buf[36], '; queries with > 3 scans', buf[37], '; max candidate chunks of a query', buf[38], '; queries above 20000 ticks', buf[39])
    print('f64 distance iteration: scans', buf[6], 'candidate chunks', buf[5])
    for name, b in (('coarse', buf[0:16]), ('fine', buf[16:32])):
        print(f'{name}: waves {b[0]} mean wave rounds {b[1] / max(b[0], 1):.1f}; queries {b[2]} mean query rounds {b[3] / max(b[2], 1):.1f}; unfinished {b[4]}; rounds histogram (x8) {list(b[8:16])}; candidate chunks per query round {b[5] / max(b[3], 1):.1f}; global scans per query {b[6] / max(b[2], 1):.2f}; f64 distance-iteration rounds {b[7]}')
except AttributeError:
    pass
