"""Row C0: Fibonacci-sphere cameras + NDC transform.  (oracle -- test infrastructure)

Follows /root/reference/utils/camera_utils.py:86-102 (fibonacci_sphere), :104-114
(calculate_up_vector), :116-245 (create_cameras).  The reference builds kaolin 0.15.0
`Camera.from_args(eye, at, up, fov=pi/4, width=height=res)` objects and only ever calls
`cam.transform(pts)`; kaolin is un-vendored (PARITY UNPINNED), so the transform below
restates the documented pinhole model: look-at view matrix, vertical fov, OpenGL-style
NDC with near=1e-2, far=1e2, depth increasing with distance.

Arithmetic contract (shared bit-for-bit with the HIP kernels, which are compiled with
-ffp-contract=off): all float32, one rounding per operation, in exactly this order
    xc = ((R00*x + R01*y) + R02*z) + t0            (same for yc, zc)
    w  = -zc
    xn = (fx*xc)/w ; yn = (fy*yc)/w ; zn = (A*zc + B)/w
Camera parameters travel as 16 float32: R (9, row-major), t (3), fx, fy, A, B.
"""
import math
import numpy as np

F32 = np.float32


def fibonacci_sphere(samples, radius):
    # camera_utils.py:86-102
    points = []
    phi = math.pi * (3. - math.sqrt(5.))
    for i in range(samples):
        y = 1 - (i / float(samples - 1)) * 2
        radius_y = math.sqrt(1 - y * y)
        theta = phi * i
        x = math.cos(theta) * radius_y * radius
        z = math.sin(theta) * radius_y * radius
        points.append((x, y * radius, z))
    return np.array(points)


def calculate_up_vector(eye_position, target_position, world_up=None):
    # camera_utils.py:104-114
    gaze = target_position - eye_position
    if world_up is None:
        world_up = np.array([0, 1, 0])
    if np.allclose(np.cross(gaze, world_up), 0):
        return np.array([0.0, 0.0, 1.0])
    side = np.cross(gaze, world_up)
    up = np.cross(side, gaze)
    return up / np.linalg.norm(up)


def camera_params(eye, at, up, fov=math.pi / 4, near=1e-2, far=1e2):
    """16 float32 camera parameters (float64 setup -> float32, as the reference's numpy->torch)."""
    eye = np.asarray(eye, np.float64)
    at = np.asarray(at, np.float64)
    up = np.asarray(up, np.float64)
    backward = eye - at
    backward = backward / np.linalg.norm(backward)
    right = np.cross(up, backward)
    right = right / np.linalg.norm(right)
    up2 = np.cross(backward, right)
    R = np.stack([right, up2, backward], 0)            # world -> camera (camera looks down -z)
    t = -R @ eye
    f = 1.0 / math.tan(fov / 2.0)
    A = -(far + near) / (far - near)
    B = -2.0 * far * near / (far - near)
    return np.concatenate([R.reshape(9), t, [f, f, A, B]]).astype(F32)


class Camera:
    """Stand-in for the kaolin Camera: exposes .transform(pts[M,3]) -> [M,3] NDC, .height, .width."""

    def __init__(self, params, res):
        self.params = np.asarray(params, F32).reshape(16)
        self.height = self.width = int(res)

    def transform(self, pts):
        return transform_points(self.params, pts)


def transform_points(params, pts):
    p = np.asarray(params, F32)
    pts = np.asarray(pts, F32)
    x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]
    cam = []
    for r in range(3):
        cam.append(((p[3 * r + 0] * x + p[3 * r + 1] * y) + p[3 * r + 2] * z) + p[9 + r])
    xc, yc, zc = cam
    w = -zc
    with np.errstate(divide='ignore', invalid='ignore'):
        xn = (p[12] * xc) / w
        yn = (p[13] * yc) / w
        zn = (p[14] * zc + p[15]) / w
    return np.stack([xn, yn, zn], -1).astype(F32)


def create_cameras(num_views=8, distance=1.6, res=512):
    """camera_utils.py:116-245, 'fibonacci_sphere' distribution.
    Returns cams, base_dirs[V,3] f32, eye_positions[V,3] f64, up_dirs[V,3] f32."""
    eyes = fibonacci_sphere(num_views, distance)
    cams, base_dirs, up_dirs = [], [], []
    at = np.array([0, 0, 0])
    for eye in eyes:
        up = calculate_up_vector(eye, at)
        cams.append(Camera(camera_params(eye, at, up), res))
        base_dirs.append((eye - at).astype(F32))
        up_dirs.append(up.astype(F32))
    return cams, np.stack(base_dirs), eyes, np.stack(up_dirs)
