"""Rows I0, Uq5: nearest-site fill.  (oracle -- test infrastructure)

Follows /root/reference/pointdreamer/ours_utils.py:610-643 (naive_inpainting, method='nearest',
scipy.interpolate.griddata -> cKDTree nearest) and /root/reference/pointdreamer/unproject.py:480-504
(dilate_atlas = the same at atlas resolution with sites = chart mask).

Every pixel takes the value of its Euclidean-nearest site.  scipy's tie-breaking among equidistant
sites is a property of its kd-tree traversal (unpinned); the build's rule is: among all sites at
the minimal squared distance, the lexicographically smallest (row, col).  At non-tie pixels this is
identical to the reference; at tie pixels tests assert the minimal-distance property instead.
"""
import numpy as np

F32 = np.float32


def nearest_site_index(site_mask):
    """site_mask[H,W] bool -> (row', col')[H,W] int64 of the nearest site under the build's rule.
    Exact separable EDT: column pass (nearest site row per column, ties -> smaller row), then per
    row an exhaustive scan over columns with composite key (dist2, row', col')."""
    m = np.asarray(site_mask, bool)
    H, W = m.shape
    if not m.any():
        raise ValueError("no sites")
    rows = np.arange(H, dtype=np.int64)[:, None]
    BIG = np.int64(1 << 30)
    up = np.where(m, rows, -BIG)
    up = np.maximum.accumulate(up, 0)                       # largest site row <= r
    down = np.where(m, rows, BIG)
    down = np.minimum.accumulate(down[::-1], 0)[::-1]       # smallest site row >= r
    d_up = rows - up
    d_down = down - rows
    use_up = d_up <= d_down
    near_row = np.where(use_up, up, down)                   # [H,W]
    g = np.minimum(d_up, d_down)                            # vertical distance (>= BIG-ish if none)
    has = g < (BIG // 2)
    cols = np.arange(W, dtype=np.int64)
    dx2 = (cols[:, None] - cols[None, :]) ** 2              # [c, c']
    out_r = np.zeros((H, W), np.int64)
    out_c = np.zeros((H, W), np.int64)
    for r in range(H):
        gr = np.where(has[r], g[r], 0)
        d2 = dx2 + (gr * gr)[None, :]
        key = (d2 * (H * W) + (near_row[r] * W + cols)[None, :])
        key = np.where(has[r][None, :], key, np.iinfo(np.int64).max)
        cbest = key.argmin(1)
        out_c[r] = cbest
        out_r[r] = near_row[r][cbest]
    return out_r, out_c


def nearest_inpaint(img, no_need_inpaint_mask2):
    """ours_utils.py:610-643 with method='nearest'.  img[C,H,W]; mask2[C,H,W] or [1,H,W] (channel 0 used);
    returns [C,H,W] float32 (the reference returns float64 copies of the same values)."""
    img = np.asarray(img, F32)
    sites = np.asarray(no_need_inpaint_mask2)[0].astype(bool)
    rr, cc = nearest_site_index(sites)
    return img[:, rr, cc]


def dilate_atlas(atlas_img, mask):
    """unproject.py:480-504.  atlas_img[A,A,3] f32, mask[1,A,A,1] bool -> [A,A,3]."""
    a = np.asarray(atlas_img, F32).transpose(2, 0, 1)
    out = nearest_inpaint(a, np.asarray(mask)[..., 0])
    return out.transpose(1, 2, 0)


def reference_nearest_inpaint_scipy(img, no_need_inpaint_mask2):
    """The reference's own formulation (scipy griddata nearest); used for the CPU baseline timing
    and to cross-check non-tie pixels.  ours_utils.py:617-643."""
    from scipy.interpolate import griddata
    img = np.asarray(img)
    m = np.asarray(no_need_inpaint_mask2)[0]
    res = img.shape[1]
    need = ~(m.astype(np.bool_))
    y_coords, x_coords = np.indices(img.shape[1:])
    coords = np.column_stack((x_coords.ravel(), y_coords.ravel()))
    img_flat = img.reshape(img.shape[0], -1)
    mask_flat = need.ravel().astype(np.bool_)
    valid_pixels = img_flat[:, ~mask_flat]
    valid_coords = coords[~mask_flat]
    xx, yy = np.meshgrid(np.arange(img.shape[2]), np.arange(res), indexing='xy')
    out = griddata(valid_coords, valid_pixels.T, (xx, yy), method='nearest')
    return out.transpose(2, 0, 1)


def reference_linear_inpaint_scipy(img, no_need_inpaint_mask2):
    """ours_utils.py:617-643 with method='linear': scipy.interpolate.griddata -> qhull Delaunay + LinearNDInterpolator
    (float64 barycentric interpolation, NaN outside the hull).  Returns [C,H,W] float64 like the reference."""
    from scipy.interpolate import griddata
    img = np.asarray(img)
    m = np.asarray(no_need_inpaint_mask2)[0]
    need = ~(m.astype(np.bool_))
    y_coords, x_coords = np.indices(img.shape[1:])
    coords = np.column_stack((x_coords.ravel(), y_coords.ravel()))
    img_flat = img.reshape(img.shape[0], -1)
    mask_flat = need.ravel().astype(np.bool_)
    xx, yy = np.meshgrid(np.arange(img.shape[2]), np.arange(img.shape[1]), indexing='xy')
    out = griddata(coords[~mask_flat], img_flat[:, ~mask_flat].T, (xx, yy), method='linear')
    return out.transpose(2, 0, 1)


def delaunay_triangle_is_valid(sites_xy, tri, q):
    """Is (sites_xy[tri]) a triangle of SOME Delaunay triangulation of the sites that contains pixel q?  Exact integer
    arithmetic: q inside or on the triangle, no site strictly inside its circumcircle.  (Pixels are a degenerate input --
    co-circular sites everywhere -- so several triangulations are valid; qhull's pick is a property of its merge order.)"""
    P = np.asarray(sites_xy, np.int64)
    a, b, c = (P[int(t)] for t in tri)
    q = np.asarray(q, np.int64)

    def orient(u, v, w):
        return (v[0] - u[0]) * (w[1] - u[1]) - (v[1] - u[1]) * (w[0] - u[0])
    o = orient(a, b, c)
    if o == 0:
        return False
    if o < 0:
        b, c = c, b
    if orient(a, b, q) < 0 or orient(b, c, q) < 0 or orient(c, a, q) < 0:
        return False
    ax, ay = a[0] - P[:, 0], a[1] - P[:, 1]
    bx, by = b[0] - P[:, 0], b[1] - P[:, 1]
    cx, cy = c[0] - P[:, 0], c[1] - P[:, 1]
    det = (ax * ax + ay * ay) * (bx * cy - cx * by) - (bx * bx + by * by) * (ax * cy - cx * ay) + (cx * cx + cy * cy) * (ax * by - bx * ay)
    return bool((det <= 0).all())
