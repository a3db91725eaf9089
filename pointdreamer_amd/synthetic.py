"""Deterministic synthetic shapes for tests and bench.

The reference ships point clouds but neither a mesh nor a UV atlas for them, and POCO / xatlas
cannot run offline, so the benchmark geometry is a stand-in: a lat-long UV sphere (radius 0.5,
50 stacks x 100 slices = 9 800 triangles, 4 902 vertices), an analytic lat-long atlas
(`gb_pos`, `mask`, `per_atlas_pixel_face_id` in the wire format of
/root/reference/demo.py:445-448), and N colored points uniform on the sphere.
Plain numpy; no reference code involved.
"""
import math
import numpy as np

F32 = np.float32


def uv_sphere(stacks=50, slices=100, radius=0.5):
    """Returns vertices[Vn,3] f32, faces[F,3] int64, face (stack, slice, half) bookkeeping."""
    verts = [(0.0, radius, 0.0)]
    for s in range(1, stacks):
        lat = math.pi * s / stacks
        y = radius * math.cos(lat)
        rr = radius * math.sin(lat)
        for k in range(slices):
            lon = 2 * math.pi * k / slices
            verts.append((rr * math.cos(lon), y, rr * math.sin(lon)))
    verts.append((0.0, -radius, 0.0))
    south = len(verts) - 1

    def vid(s, k):
        return 1 + (s - 1) * slices + (k % slices)
    faces = []
    face_of_quad = {}
    for k in range(slices):                      # north fan (stack 0)
        face_of_quad[(0, k, 0)] = len(faces)
        face_of_quad[(0, k, 1)] = len(faces)
        faces.append((0, vid(1, k + 1), vid(1, k)))
    for s in range(1, stacks - 1):
        for k in range(slices):
            a, b, c, d = vid(s, k), vid(s, k + 1), vid(s + 1, k), vid(s + 1, k + 1)
            face_of_quad[(s, k, 0)] = len(faces)
            faces.append((a, b, c))
            face_of_quad[(s, k, 1)] = len(faces)
            faces.append((b, d, c))
    for k in range(slices):                      # south fan
        face_of_quad[(stacks - 1, k, 0)] = len(faces)
        face_of_quad[(stacks - 1, k, 1)] = len(faces)
        faces.append((south, vid(stacks - 1, k), vid(stacks - 1, k + 1)))
    lut = np.zeros((stacks, slices, 2), np.int64)
    for (s, k, h), f in face_of_quad.items():
        lut[s, k, h] = f
    return np.array(verts, F32), np.array(faces, np.int64), lut


def uv_sphere_uvs(stacks=50, slices=100, A=1024, gutter=2):
    """Explicit UVs of the lat-long sphere in the layout `latlong_atlas` rasterises (what xatlas would hand over as `uvs`,
    `mesh_tex_idx`): uvs[(stacks+1)*(slices+1), 2] f32 in [0,1] (u = column / A, v = row / A; seam column duplicated, one
    pole UV per fan triangle), face_uv_idx[F,3] int64 aligned with `uv_sphere`'s faces."""
    span = A - 2 * gutter
    uv = np.zeros(((stacks + 1) * (slices + 1), 2), np.float64)
    for s in range(stacks + 1):
        for k in range(slices + 1):
            kf = (k + 0.5) / slices if s in (0, stacks) else k / slices       # poles: one UV per fan triangle
            uv[s * (slices + 1) + k] = ((gutter + min(kf, 1.0) * span) / A, (gutter + s / stacks * span) / A)

    def uid(s, k):
        return s * (slices + 1) + k
    fuv = []
    for k in range(slices):
        fuv.append((uid(0, k), uid(1, k + 1), uid(1, k)))
    for s in range(1, stacks - 1):
        for k in range(slices):
            fuv.append((uid(s, k), uid(s, k + 1), uid(s + 1, k)))
            fuv.append((uid(s, k + 1), uid(s + 1, k + 1), uid(s + 1, k)))
    for k in range(slices):
        fuv.append((uid(stacks, k), uid(stacks - 1, k), uid(stacks - 1, k + 1)))
    return uv.astype(F32), np.array(fuv, np.int64)


def face_normals(vertices, faces):
    """Unit face normals (what kal.ops.mesh.face_normals(unit=True) provides at demo.py:422)."""
    v = vertices.astype(np.float64)
    n = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    return n.astype(F32)


def latlong_atlas(A, stacks=50, slices=100, radius=0.5, gutter=2, n_charts=1, lut=None):
    """Analytic atlas: texel (row i, col j) -> (v, u) -> (lat, lon) on the sphere.
    n_charts > 1 splits the u range into strips separated by `gutter`-texel background gaps.
    Returns gb_pos[1,A,A,3] f32, mask[1,A,A,1] bool, per_atlas_pixel_face_id[1,A,A] int64 (-1 background)."""
    ii, jj = np.meshgrid(np.arange(A), np.arange(A), indexing='ij')
    mask = (ii >= gutter) & (ii < A - gutter) & (jj >= gutter) & (jj < A - gutter)
    if n_charts > 1:
        w = A // n_charts
        for c in range(1, n_charts):
            mask &= ~((jj >= c * w - gutter) & (jj < c * w + gutter))
    span = A - 2 * gutter
    v = (ii - gutter + 0.5) / span
    u = (jj - gutter + 0.5) / span
    lat = np.pi * np.clip(v, 0, 1)
    lon = 2 * np.pi * np.clip(u, 0, 1)
    pos = np.stack([radius * np.sin(lat) * np.cos(lon), radius * np.cos(lat), radius * np.sin(lat) * np.sin(lon)], -1)
    s = np.clip((np.clip(v, 0, 1) * stacks).astype(np.int64), 0, stacks - 1)
    k = np.clip((np.clip(u, 0, 1) * slices).astype(np.int64), 0, slices - 1)
    fs = np.clip(v, 0, 1) * stacks - s
    fk = np.clip(u, 0, 1) * slices - k
    half = ((fs + fk) > 1.0).astype(np.int64)
    if lut is None:
        _, _, lut = uv_sphere(stacks, slices, radius)
    fid = lut[s, k, half]
    fid = np.where(mask, fid, -1)
    gb_pos = np.where(mask[..., None], pos, 0.0).astype(F32)
    return gb_pos[None], mask[None, :, :, None], fid[None]


def sphere_points(n, radius=0.5, seed=0, noise=0.05):
    """n points uniform on the sphere + smooth-plus-noise colours in [0,1]."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, 3))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    xyz = (radius * g).astype(F32)
    x, y = g[:, 0], g[:, 1]
    cols = []
    for c in range(3):
        ph = 2.0 * c
        cols.append(0.5 + 0.45 * np.sin(6 * np.pi * x * 0.5 + ph) * np.cos(4 * np.pi * y * 0.5 + 0.5 * ph))
    rgb = np.stack(cols, 1) + rng.uniform(-noise, noise, (n, 3))
    return xyz, np.clip(rgb, 0, 1).astype(F32)


def make_shape(n_points=30000, A=1024, stacks=50, slices=100, seed=0, n_charts=1, gutter=2):
    """One full synthetic shape in the reference's tensor contracts (numpy)."""
    vertices, faces, lut = uv_sphere(stacks, slices)
    gb_pos, mask, fid = latlong_atlas(A, stacks, slices, gutter=gutter, n_charts=n_charts, lut=lut)
    xyz, rgb = sphere_points(n_points, seed=seed)
    return dict(vertices=vertices, faces=faces, f_normals=face_normals(vertices, faces),
                gb_pos=gb_pos, mask=mask, per_atlas_pixel_face_id=fid, points=xyz, colors=rgb)
