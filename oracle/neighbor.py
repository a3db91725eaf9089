"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of SURVEY 8(f)-2, `complete_unseen_by='neighbor'`:
  paint_invisible_areas_by_neighbors   /root/reference/pointdreamer/unproject.py:93-196
  compute_vertex_only_uv_mask          /root/reference/pointdreamer/unproject.py:17-37
  subdivide_with_uv                    /root/reference/utils/mesh_utils.py:7-114
Third-party pieces the reference calls and that are not vendored (restated from their published behaviour):
  trimesh 4.x  grouping.unique_rows (rows bit-packed into one int64, np.unique with return_index / return_inverse) and
               geometry.faces_to_edges (edges 0-1, 1-2, 2-0 per face);
  kaolin 0.15  ops.mesh.uniform_laplacian (1/deg on the unique neighbours, -1 diagonal, isolated rows 0).
Pinned against the imported reference by tools/gen_golden_neighbor.py -> tests/golden/neighbor_*.npz.

Deterministic rules where the reference is order-dependent (GPU index_put with duplicate indices): a vertex that owns several
UVs takes the one with the LARGEST uv index (= torch CPU's last-write-wins over the lexicographically sorted unique pairs);
several subdivided vertices landing on one texel: the LARGEST vertex index wins.  The neighbour average accumulates in
float32 over ascending neighbour index (the reference's dense sgemm order is unspecified: parity to 1e-5).
"""
import numpy as np

from . import inpaint as oinp

F32 = np.float32


def unique_rows(data):
    d = np.asarray(data).astype(np.int64)
    prec = 64 // d.shape[1]
    h = np.zeros(len(d), np.int64)
    for off, col in enumerate(d.T):
        h ^= col << (off * prec)
    _, unique, inverse = np.unique(h, return_index=True, return_inverse=True)
    return unique, inverse


def faces_to_edges(faces):
    return np.asarray(faces)[:, [0, 1, 1, 2, 2, 0]].reshape((-1, 2))


def subdivide_with_uv(vertices, faces, face_uv_idx, uvs, face_index=None):
    """mesh_utils.py:7-114: midpoint subdivision of the faces in `face_index` (their neighbours are left alone)."""
    face_mask = np.ones(len(faces), bool) if face_index is None else np.zeros(len(faces), bool)
    if face_index is not None:
        face_mask[face_index] = True
    fs, fu = faces[face_mask], face_uv_idx[face_mask]
    edges = np.sort(faces_to_edges(fs), axis=1)
    unique, inverse = unique_rows(edges)
    mid = vertices[edges[unique]].mean(axis=1)
    mid_idx = inverse.reshape((-1, 3)) + len(vertices)
    edges_uv = np.sort(faces_to_edges(fu), axis=1)
    unique_uv, inverse_uv = unique_rows(edges_uv)
    mid_uv = uvs[edges_uv[unique_uv]].mean(axis=1)
    mid_idx_uv = inverse_uv.reshape((-1, 3)) + len(uvs)

    def split(f, m):
        return np.column_stack([f[:, 0], m[:, 0], m[:, 2], m[:, 0], f[:, 1], m[:, 1], m[:, 2], m[:, 1], f[:, 2],
                                m[:, 0], m[:, 1], m[:, 2]]).reshape((-1, 3))
    new_faces = np.vstack((faces[~face_mask], split(fs, mid_idx)))
    new_face_uv_idx = np.vstack((face_uv_idx[~face_mask], split(fu, mid_idx_uv)))
    return np.vstack((vertices, mid)), new_faces, np.vstack((uvs, mid_uv)), new_face_uv_idx


def subdivide_twice(vertices, faces, uvs, face_uv_idx, to_inpaint_face_id):
    """unproject.py:106-114: two rounds with the SAME face index list (indices are not re-mapped after round one -- kept)."""
    v, f, u, fu = vertices, faces, uvs, face_uv_idx
    for _ in range(2):
        v, f, u, fu = subdivide_with_uv(v, f, fu, u, face_index=to_inpaint_face_id)
    return v, f, u, fu


def vertex_uvs(num_vertices, faces, face_uv_idx, uvs):
    """unproject.py:123-127: one UV per vertex out of the unique (vertex, uv) pairs; last write wins."""
    pairs = np.unique(np.stack([faces.reshape(-1), face_uv_idx.reshape(-1)], 1), axis=0)
    out = np.zeros((num_vertices, 2), F32)
    out[pairs[:, 0]] = uvs[pairs[:, 1]].astype(F32)          # numpy fancy assignment: last occurrence wins, as torch CPU
    return out


def vertex_texels(vert_uvs, res):
    """unproject.py:130-134: (row, col) = long(clip(uv * res, 0, res-1))[..., (1, 0)]."""
    pix = np.clip((vert_uvs.astype(F32) * F32(res)), 0, res - 1).astype(np.int64)
    return np.stack([pix[:, 1], pix[:, 0]], 1)


def adjacency_csr(num_vertices, faces):
    """Unique undirected neighbours (kaolin adjacency_matrix): CSR rowptr[V+1], colidx sorted ascending per row."""
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    e = np.concatenate([e, e[:, ::-1]], 0)
    e = e[e[:, 0] != e[:, 1]]
    e = np.unique(e, axis=0)
    rowptr = np.zeros(num_vertices + 1, np.int64)
    np.add.at(rowptr, e[:, 0] + 1, 1)
    return np.cumsum(rowptr), e[:, 1].copy()


def diffuse(colors, has_color, rowptr, colidx, max_rounds=10000):
    """unproject.py:139-181.  Jacobi iteration over the invalid vertices: new = (sum_j w c_j n_j) / (sum_j w n_j) with
    w = 1/deg(i) over the neighbours (L + I has a zero diagonal), applied where the denominator is > 0; loop control as the
    reference: rounds that colour new vertices count up, then the same number of rounds is run again without progress.
    Returns (colors[V,3] f32, colored[V] bool, iterations)."""
    colors = colors.astype(F32).copy()
    count = has_color.astype(F32).copy()
    invalid = np.nonzero(~has_color)[0]
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.int64)
    total = count.sum(dtype=np.float64)
    rounds, stage, iters = 0, "uncolored", 0
    while stage == "uncolored" or rounds > 0:
        new_c = np.zeros((len(invalid), 3), F32)
        new_n = np.zeros((len(invalid),), F32)
        for a, i in enumerate(invalid):
            if deg[i] == 0:
                continue
            w = F32(1.0) / F32(deg[i])
            accc = np.zeros(3, F32)
            accn = F32(0.0)
            for j in colidx[rowptr[i]:rowptr[i + 1]]:
                accc = accc + w * (colors[j] * count[j])
                accn = accn + w * count[j]
            new_c[a], new_n[a] = accc, accn
        upd = new_n > 0
        colors[invalid[upd]] = new_c[upd] / new_n[upd, None]
        count[invalid] = upd.astype(F32)
        new_total = count.sum(dtype=np.float64)
        if new_total > total:
            total = new_total
            rounds += 1
        else:
            stage = "colored"
            rounds -= 1
        iters += 1
        if rounds > max_rounds:
            break
    return colors, count > 0, iters


def paint_invisible_areas_by_neighbors(vertices, faces, uvs, face_uv_idx, to_inpaint_face_id, atlas_img, atlas_inpainted_mask,
                                       return_intermediates=False):
    """unproject.py:93-196 with use_atlas=True.  atlas_img [A,A,3] f32, atlas_inpainted_mask [A,A] bool -> atlas [A,A,3]."""
    res = atlas_inpainted_mask.shape[1]
    sv, sf, su, sfu = subdivide_twice(np.asarray(vertices), np.asarray(faces), np.asarray(uvs), np.asarray(face_uv_idx),
                                      np.asarray(to_inpaint_face_id))
    vuv = vertex_uvs(len(sv), sf, sfu, su)
    tex = vertex_texels(vuv, res)
    colors = atlas_img[tex[:, 0], tex[:, 1]].astype(F32)
    has = atlas_inpainted_mask[tex[:, 0], tex[:, 1]].astype(bool)
    rowptr, colidx = adjacency_csr(len(sv), sf)
    colors, colored, iters = diffuse(colors, has, rowptr, colidx)
    atlas = atlas_img.astype(F32).copy()
    mask = atlas_inpainted_mask.astype(bool).copy()
    atlas[tex[:, 0], tex[:, 1]] = colors                       # last (largest vertex index) wins
    mask[tex[:, 0], tex[:, 1]] = True
    filled = oinp.nearest_inpaint(atlas.transpose(2, 0, 1), mask[None].astype(F32)).transpose(1, 2, 0)
    if return_intermediates:
        return dict(atlas=filled, atlas_before_fill=atlas, mask_before_fill=mask, vertices=sv, faces=sf, uvs=su, face_uv_idx=sfu,
                    texels=tex, vert_colors=colors, colored=colored, iterations=iters)
    return filled
