import os, sys, time, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pointdreamer_amd import demo, synthetic, io_utils
print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None, 'torch threads', torch.get_num_threads())
tmp = tempfile.mkdtemp()
xyz, rgb = synthetic.sphere_points(30000, seed=3)
pc = os.path.join(tmp, 'ball.ply'); io_utils.save_colored_pc_ply(xyz, rgb, pc)
dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
def T(label, f):
    torch.cuda.synchronize(); t = time.time(); r = f(); torch.cuda.synchronize(); print(f'{label:40s} {(time.time()-t)*1e3:8.2f} ms'); return r
for rep in range(3):
    print('--- rep', rep)
    x, c = T('read_ply', lambda: io_utils.read_ply_xyzrgb(pc))
    a = T('np.asarray f32', lambda: np.asarray(x, np.float32))
    tt = T('torch.tensor(xyz)', lambda: torch.tensor(a))
    td = T('.to(device)', lambda: tt.to(dev))
    tc = T('torch.tensor(rgb).float()', lambda: torch.tensor(np.asarray(c)).float())
    T('min/max', lambda: (td.min(0)[0], td.max(0)[0]))
    T('save_colored_pc_ply', lambda: io_utils.save_colored_pc_ply(td.cpu().numpy(), tc.numpy(), os.path.join(tmp, 'o.ply')))
    import logging
    v, f, d = T('standin_geometry', lambda: demo.standin_geometry(td, 1024, dev, logging.getLogger('x')))
    T('face_normals', lambda: torch.from_numpy(synthetic.face_normals(v.cpu().numpy(), f.cpu().numpy())).to(dev))
