"""Golden vectors of round 3, produced by RUNNING THE REFERENCE (build container only):  python -m tools.gen_golden_r3
  camera_distributions.npz   utils/camera_utils.create_cameras (camera_utils.py:116-245) for every camera distribution: eye
                             positions, base directions, up vectors and the fov handed to the (stubbed) kaolin camera."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import ref_harness as rh            # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    rh.install()
    import importlib
    cu = importlib.import_module('utils.camera_utils')
    seen = []

    class Cam:
        @staticmethod
        def from_args(**kw):
            seen.append(float(kw['fov']))
            return None
    cu.kal.render.camera.Camera = Cam
    out = {}
    for tag, kw in (('fib8', dict(num_views=8, distance=1.6, distribution='fibonacci_sphere')),
                    ('fib5', dict(num_views=5, distance=2.0, distribution='fibonacci_sphere')),
                    ('self6', dict(num_views=6, distance=1.6, distribution='self_defined')),
                    ('self20', dict(num_views=20, distance=1.6, distribution='self_defined')),
                    ('blender', dict(num_views=8, distance=1.6, distribution='blender')),
                    ('exact_blender', dict(num_views=3, distance=1.0, distribution='exact_blender'))):
        del seen[:]
        cams, base_dirs, eyes, ups = cu.create_cameras(res=64, device=torch.device('cpu'), **kw)
        out[tag + '_eyes'] = np.asarray(eyes, np.float64)
        out[tag + '_base_dirs'] = base_dirs.numpy()
        out[tag + '_up_dirs'] = ups.numpy()
        out[tag + '_fov'] = np.array(seen, np.float64)
        out[tag + '_args'] = np.array([kw['num_views'], kw['distance']], np.float64)
    np.savez_compressed(os.path.join(OUT, 'camera_distributions.npz'), **out)
    print('camera_distributions.npz', sorted(out)[:6], '...')


if __name__ == '__main__':
    main()
