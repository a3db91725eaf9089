"""Cameras for the texturing path (row C0).  Mirrors the interface of the reference's
utils/camera_utils.py:86-245 (`fibonacci_sphere`, `calculate_up_vector`, `create_cameras`) but the
camera object is a plain 16-float struct consumed by the HIP kernels instead of a kaolin Camera:
R (9, row-major world->camera), t (3), fx, fy, A, B  with  NDC = (fx*xc/-zc, fy*yc/-zc, (A*zc+B)/-zc),
vertical fov pi/4, near 1e-2, far 1e2 (kaolin's defaults; kaolin itself is not a dependency).
"""
import math
import numpy as np
import torch


def fibonacci_sphere(samples, radius):
    pts = []
    phi = math.pi * (3. - math.sqrt(5.))
    for i in range(samples):
        y = 1 - (i / float(samples - 1)) * 2
        ry = math.sqrt(1 - y * y)
        th = phi * i
        pts.append((math.cos(th) * ry * radius, y * radius, math.sin(th) * ry * radius))
    return np.array(pts)


def calculate_up_vector(eye_position, target_position, world_up=None):
    gaze = target_position - eye_position
    if world_up is None:
        world_up = np.array([0, 1, 0])
    if np.allclose(np.cross(gaze, world_up), 0):
        return np.array([0.0, 0.0, 1.0])
    side = np.cross(gaze, world_up)
    up = np.cross(side, gaze)
    return up / np.linalg.norm(up)


def look_at_params(eye, at, up, fov=math.pi / 4, near=1e-2, far=1e2):
    eye, at, up = (np.asarray(a, np.float64) for a in (eye, at, up))
    back = eye - at
    back /= np.linalg.norm(back)
    right = np.cross(up, back)
    right /= np.linalg.norm(right)
    up2 = np.cross(back, right)
    R = np.stack([right, up2, back], 0)
    t = -R @ eye
    f = 1.0 / math.tan(fov / 2.0)
    return np.concatenate([R.reshape(9), t, [f, f, -(far + near) / (far - near), -2.0 * far * near / (far - near)]]
                          ).astype(np.float32)


class Camera:
    """Drop-in for the kaolin Camera as far as the hot path uses it: .transform, .height, .width."""

    def __init__(self, params, res, device):
        self.params = torch.as_tensor(params, dtype=torch.float32).reshape(16).to(device).contiguous()
        self.height = self.width = int(res)

    def transform(self, pts):
        from .ours_utils import transform_points
        return transform_points([self], pts)[0]


def stack_params(cams):
    from . import _lib
    return _lib.memo([c.params for c in cams], 'stack_params', lambda ps: torch.stack(list(ps), 0).contiguous())


def create_cameras(num_views=8, distance=1.6, res=512, distribution='fibonacci_sphere',
                   device=torch.device('cuda'), vis=False):
    """Same return contract as the reference: cams, base_dirs[V,3], eye_positions (numpy), up_dirs[V,3]."""
    if distribution != 'fibonacci_sphere':
        raise NotImplementedError("only camera_distribution='fibonacci_sphere' (every shipped config) is built")
    eyes = fibonacci_sphere(num_views, distance)
    cams = []
    base_dirs = torch.zeros((num_views, 3), dtype=torch.float32)
    up_dirs = torch.zeros((num_views, 3), dtype=torch.float32)
    at = np.array([0, 0, 0])
    for i, eye in enumerate(eyes):
        up = calculate_up_vector(eye, at)
        cams.append(Camera(look_at_params(eye, at, up), res, device))
        base_dirs[i] = torch.tensor(eye - at).float()
        up_dirs[i] = torch.tensor(up).float()
    return cams, base_dirs.to(device), eyes, up_dirs.to(device)
