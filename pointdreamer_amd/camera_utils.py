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


def _dodecahedron():
    phi = (1 + math.sqrt(5)) / 2.
    return np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1],
                     [0, -phi, -1 / phi], [0, -phi, 1 / phi], [0, phi, -1 / phi], [0, phi, 1 / phi],
                     [-1 / phi, 0, -phi], [-1 / phi, 0, phi], [1 / phi, 0, -phi], [1 / phi, 0, phi],
                     [-phi, -1 / phi, 0], [-phi, 1 / phi, 0], [phi, -1 / phi, 0], [phi, 1 / phi, 0]]).astype(float)


def create_cameras(num_views=8, distance=1.6, res=512, distribution='fibonacci_sphere',
                   device=torch.device('cuda'), vis=False):
    """Same return contract as the reference: cams, base_dirs[V,3], eye_positions (numpy), up_dirs[V,3]."""
    if distribution not in ('fibonacci_sphere', 'self_defined', 'blender', 'exact_blender'):
        raise ValueError(f"camera distribution {distribution!r} (camera_utils.py:129: fibonacci_sphere, self_defined, blender, exact_blender)")
    fov = math.pi * 45 / 180
    if distribution == 'fibonacci_sphere':
        eyes = fibonacci_sphere(num_views, distance)
    elif distribution in ('blender', 'exact_blender'):
        # camera_utils.py:132-164: the 20 vertices of a dodecahedron, 1.2 x its circumradius away, y-up -> z-up; always 20 views
        num_views = 20
        eyes = (_dodecahedron() * 1.2).dot(np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0.0]]).T)
        if distribution == 'exact_blender':
            fov = 0.8575560450553894
    else:
        # camera_utils.py:165-201: six axis views at `distance`, or the raw dodecahedron vertices
        if num_views == 6:
            eyes = distance * np.array([[0, 0, -1.0], [0, 0, 1.0], [0, -1.0, 0], [0, 1.0, 0], [-1.0, 0, 0], [1.0, 0, 0]])
        elif num_views == 20:
            eyes = _dodecahedron()
        else:
            raise ValueError("camera distribution 'self_defined' knows 6 or 20 views (camera_utils.py:165-201)")
    cams = []
    base_dirs = torch.zeros((num_views, 3), dtype=torch.float32)
    up_dirs = torch.zeros((num_views, 3), dtype=torch.float32)
    at = np.array([0, 0, 0])
    for i, eye in enumerate(eyes):
        up = calculate_up_vector(eye, at)
        cams.append(Camera(look_at_params(eye, at, up, fov=fov), res, device))
        base_dirs[i] = torch.tensor(eye - at).float()
        up_dirs[i] = torch.tensor(up).float()
    return cams, base_dirs.to(device), eyes, up_dirs.to(device)


def get_cam_Ks_RTs_from_locations(cam_locations):
    """camera_utils.py:940-985: world -> camera [R | t] (rows U, V, N = right, up, look direction) of cameras at `cam_locations`
    [V,3] looking at the origin, y-up (z-up when the view direction is vertical), and the fixed 512-pixel intrinsics.
    Returns (cam_K [3,3], cam_RTs [V,3,4]) as float64 numpy arrays."""
    loc = cam_locations.detach().cpu().numpy() if torch.is_tensor(cam_locations) else np.asarray(cam_locations)
    loc = loc.astype(np.float64)
    cam_RTs = np.zeros((len(loc), 3, 4))
    target = np.array([0.0, 0.0, 0.0])
    for i, eye in enumerate(loc):
        N = target - eye
        N = N / np.linalg.norm(N)
        up = np.array([0.0, 0.0, 1.0]) if (N[0] == 0 and N[2] == 0) else np.array([0.0, 1.0, 0.0])
        U = np.cross(N, up)
        U = U / np.linalg.norm(U)
        V = np.cross(U, N)
        V = V / np.linalg.norm(V)
        cam_RTs[i] = np.array([[U[0], U[1], U[2], np.dot(-U, eye)],
                               [V[0], V[1], V[2], np.dot(-V, eye)],
                               [N[0], N[1], N[2], np.dot(-N, eye)]])
    cam_K = np.array([[560.0, 0, 256], [0, 560, 256], [0, 0, 1]])
    return cam_K, cam_RTs
