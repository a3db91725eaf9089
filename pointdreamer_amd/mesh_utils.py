"""Host-side mesh helpers of the `complete_unseen_by='neighbor'` path.

`subdivide_with_uv` keeps the name, argument order and return tuple of /root/reference/utils/mesh_utils.py:7-114
(numpy on the host there as well; the reference builds it on trimesh's `grouping.unique_rows` / `faces_to_edges`).
The numbering contract that downstream code depends on: new vertices (UVs) are appended after the old ones, one per unique
undirected edge of the selected faces, ordered by (larger endpoint, smaller endpoint); untouched faces come first, then
four children per selected face in the order (v0 m01 m20) (m01 v1 m12) (m20 m12 v2) (m01 m12 m20).
"""
import numpy as np


def _edge_midpoints(tri, n_existing):
    """Per-corner midpoint ids (m01, m12, m20) for triangles `tri` [T,3] and the endpoint pairs of the new points."""
    a = tri[:, [0, 1, 2]].reshape(-1)
    b = tri[:, [1, 2, 0]].reshape(-1)
    lo, hi = np.minimum(a, b).astype(np.int64), np.maximum(a, b).astype(np.int64)
    key = lo | (hi << 32)                                   # sort key: larger endpoint major, smaller endpoint minor
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    ends = np.stack([lo[first], hi[first]], 1)
    return inv.reshape(-1, 3) + n_existing, ends


def _children(tri, mid):
    v0, v1, v2 = tri[:, 0], tri[:, 1], tri[:, 2]
    m01, m12, m20 = mid[:, 0], mid[:, 1], mid[:, 2]
    kids = np.stack([np.stack([v0, m01, m20], 1), np.stack([m01, v1, m12], 1), np.stack([m20, m12, v2], 1),
                     np.stack([m01, m12, m20], 1)], 1)      # [T,4,3]
    return kids.reshape(-1, 3)


def subdivide_with_uv(vertices, faces, face_uv_idx, uvs, face_index=None):
    """Midpoint-subdivide the faces in `face_index` (all faces if None); their neighbours are left untouched, so the result
    is not watertight -- exactly what the reference does.  Returns (new_vertices, new_faces, new_uvs, new_face_uv_idx)."""
    vertices, faces, face_uv_idx, uvs = (np.asarray(x) for x in (vertices, faces, face_uv_idx, uvs))
    pick = np.zeros(len(faces), bool)
    if face_index is None:
        pick[:] = True
    else:
        pick[np.asarray(face_index)] = True
    tri, tri_uv = faces[pick], face_uv_idx[pick]
    mid, ends = _edge_midpoints(tri, len(vertices))
    mid_uv, ends_uv = _edge_midpoints(tri_uv, len(uvs))
    new_vertices = np.concatenate([vertices, vertices[ends].mean(axis=1)], 0)
    new_uvs = np.concatenate([uvs, uvs[ends_uv].mean(axis=1)], 0)
    new_faces = np.concatenate([faces[~pick], _children(tri, mid)], 0)
    new_face_uv_idx = np.concatenate([face_uv_idx[~pick], _children(tri_uv, mid_uv)], 0)
    return new_vertices, new_faces, new_uvs, new_face_uv_idx


def vertex_uv_table(num_vertices, faces, face_uv_idx, uvs):
    """One UV per vertex (unproject.py:123-127): of the UVs a vertex is used with, the one with the largest index."""
    v = np.asarray(faces).reshape(-1).astype(np.int64)
    t = np.asarray(face_uv_idx).reshape(-1).astype(np.int64)
    best = np.full(num_vertices, -1, np.int64)
    np.maximum.at(best, v, t)
    out = np.zeros((num_vertices, 2), np.float32)
    used = best >= 0
    out[used] = np.asarray(uvs, np.float32)[best[used]]
    return out


def neighbour_csr(num_vertices, faces):
    """Unique undirected vertex neighbours as CSR (int32 rowptr[V+1], colidx ascending per row)."""
    f = np.asarray(faces).astype(np.int64)
    a = np.concatenate([f[:, 0], f[:, 1], f[:, 2], f[:, 1], f[:, 2], f[:, 0]])
    b = np.concatenate([f[:, 1], f[:, 2], f[:, 0], f[:, 0], f[:, 1], f[:, 2]])
    keep = a != b
    key = np.unique(a[keep] * num_vertices + b[keep])
    rows, cols = key // num_vertices, key % num_vertices
    rowptr = np.zeros(num_vertices + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), cols.astype(np.int32)
