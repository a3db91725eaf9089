"""Row P3b: hidden-point removal (Katz et al.), the reference's Open3D call at ours_utils.py:204-225.

The reference runs this stage on the HOST (open3d -> qhull, float64, points copied to the CPU per view);
this module keeps it a host stage and drives the same qhull library through scipy.spatial.ConvexHull:
visible set = vertices of the convex hull of {spherical flip of the points} U {eye}.  It is not on the
measured hot path (SURVEY 8d reports it separately) and a device kernel is listed as "next" in DESIGN.md.
"""
import numpy as np
import torch


def hidden_point_removal(points, eye_positions, radius):
    from scipy.spatial import ConvexHull
    pts = points.detach().double().cpu().numpy()
    out = np.zeros((len(eye_positions), pts.shape[0]), bool)
    for i, eye in enumerate(eye_positions):
        q = pts - np.asarray(eye, np.float64)[None]
        n = np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-300)
        flipped = q + 2 * (radius - n) * q / n
        hull = ConvexHull(np.concatenate([flipped, np.zeros((1, 3))], 0))
        vid = hull.vertices
        out[i, vid[vid < pts.shape[0]]] = True
    return torch.from_numpy(out).to(points.device)
