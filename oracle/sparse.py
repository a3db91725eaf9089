"""Rows P4, P5, P6: sparse per-view images and masks.  (oracle -- test infrastructure)

Follows /root/reference/pointdreamer/ours_utils.py:456-495 (paint_pixels), :497-532
(get_forground_inner_edge_mask, method='dilate'), :954-1044 (get_one_sparse_img), :848-882
(get_sparse_images).

Rules fixed where the reference is order-undefined:
  * duplicate splats: the LARGEST point index wins (== torch CPU index_put_ with one thread,
    which is how the golden vectors were generated);
  * edge pixel -> nearest valid point (kaolin sided_distance, un-vendored): exact integer squared
    distance on (row,col), ties -> smallest index in the valid-point list;
  * a view with zero valid points or zero foreground (the reference divides by zero / fails in the
    NN call) returns the all-background result: sparse=0, mask0=fg, mask2=1-fg, scale_factor=1.
"""
import numpy as np
from .project import resize_mask_bilinear_nonzero

F32 = np.float32


def paint_pixels(img, pixel_coords, pixel_colors, point_size):
    """ours_utils.py:456-495.  img[C,r,r]; sequential writes => last (largest) index wins."""
    N = pixel_coords.shape[0]
    C = img.shape[0]
    if not isinstance(pixel_colors, np.ndarray):
        pixel_colors = np.full((N, C), pixel_colors, F32)
    if N == 0:
        return img
    H, W = img.shape[1:]
    if point_size == 1:
        g = pixel_coords
        cols = pixel_colors
    else:
        s = point_size
        offs = np.arange(-s + 1, s)
        xx, yy = np.meshgrid(offs, offs, indexing='ij')
        grid = np.stack([xx, yy], 2)[None] + pixel_coords[:, None, None, :]     # [N,g,g,2]
        cols = np.broadcast_to(pixel_colors[:, None, None, :], grid.shape[:3] + (C,))
        m = (grid[..., 0] >= 0) & (grid[..., 0] < H) & (grid[..., 1] >= 0) & (grid[..., 1] < W)
        g = grid[m]
        cols = cols[m]
    # the write with the largest position in write order wins (sequential index_put_)
    lin = g[:, 0] * W + g[:, 1]
    winner = np.full(H * W, -1, np.int64)
    np.maximum.at(winner, lin, np.arange(lin.shape[0]))
    hit = np.nonzero(winner >= 0)[0]
    img.reshape(C, -1)[:, hit] = cols[winner[hit]].T
    return img


def foreground_inner_edge_mask(fg):
    """ours_utils.py:519-522: maxpool3x3(~fg, pad 1 with -inf) & fg."""
    bg = ~fg
    H, W = fg.shape
    p = np.zeros((H + 2, W + 2), bool)
    p[1:-1, 1:-1] = bg
    dil = np.zeros_like(bg)
    for dy in range(3):
        for dx in range(3):
            dil |= p[dy:dy + H, dx:dx + W]
    return dil & fg


def nearest_valid_point(edge_coords, valid_coords):
    """Replaces kal.metrics.pointcloud.sided_distance at ours_utils.py:1013-1018.
    Integer squared distance, first (smallest) index among minima."""
    idx = np.zeros(edge_coords.shape[0], np.int64)
    vc = valid_coords.astype(np.int64)
    for s in range(0, edge_coords.shape[0], 256):
        e = edge_coords[s:s + 256].astype(np.int64)
        d = ((e[:, None, :] - vc[None, :, :]) ** 2).sum(-1)
        idx[s:s + 256] = d.argmin(1)
    return idx


def get_one_sparse_img(point_pixels, colors, point_validation, hard_mask, res, point_size, edge_point_size,
                       mask_ratio_thresh=0.82):
    """ours_utils.py:954-1044.  point_pixels[N,2] int64 (row,col); colors[N,3] f32; hard_mask[r,r] bool.
    Returns sparse[3,r,r], mask0[3,r,r], mask2[3,r,r] (f32, flipped vertically), mask_ratio, scale_factor (f32)."""
    point_pixels = np.asarray(point_pixels, np.int64)
    colors = np.asarray(colors, F32)
    point_validation = np.asarray(point_validation, bool)
    hard_mask = np.asarray(hard_mask, bool)
    fg_num = F32(hard_mask.sum())
    valid_num = int(point_validation.sum())
    degenerate = (valid_num == 0) or (fg_num == 0)
    scale_factor = F32(1)
    if not degenerate:
        mask_ratio = F32(1) - F32(valid_num) / fg_num
        if mask_ratio > F32(mask_ratio_thresh):
            wanted = F32(valid_num) / F32(1 - mask_ratio_thresh)
            scale_factor = F32(wanted / fg_num)
            uv = point_pixels.astype(F32) / F32(res)
            uv = uv * F32(2) - F32(1)
            uv = uv * scale_factor
            uv = (uv + F32(1)) * F32(0.5)
            pp = uv * F32(res)
            pp = np.clip(pp, F32(0), F32(res - 1))
            point_pixels = pp.astype(np.int64)
            after_res = int(np.floor(F32(res) * scale_factor))
            if (res - after_res) % 2 == 1:
                after_res += 1
            pad = int((res - after_res) / 2)
            small = resize_mask_bilinear_nonzero(hard_mask, after_res, after_res)
            hard_mask = np.zeros((res, res), bool)
            hard_mask[pad:pad + after_res, pad:pad + after_res] = small
    sparse = np.zeros((3, res, res), F32)
    mask0 = np.broadcast_to(hard_mask[None].astype(F32), (3, res, res)).copy()
    mask2 = F32(1) - mask0
    if not degenerate:
        valid_pp = point_pixels[point_validation]
        valid_col = colors[point_validation]
        sparse = paint_pixels(sparse, valid_pp, valid_col, point_size)
        edge = foreground_inner_edge_mask(hard_mask)
        edge_coords = np.argwhere(edge)                       # row-major order == torch.nonzero
        idx = nearest_valid_point(edge_coords, valid_pp)
        sparse = paint_pixels(sparse, edge_coords, valid_col[idx], edge_point_size)
        mask2 = paint_pixels(mask2, valid_pp, 1.0, point_size)
        mask2 = paint_pixels(mask2, edge_coords, 1.0, edge_point_size)
    fg2 = F32(hard_mask.sum())
    with np.errstate(divide='ignore', invalid='ignore'):
        mask_ratio = F32(1) - F32((mask2[0] * hard_mask).sum()) / fg2
    return sparse[:, ::-1].copy(), mask0[:, ::-1].copy(), mask2[:, ::-1].copy(), mask_ratio, scale_factor


def get_sparse_images(point_pixels, colors, point_validation, hard_masks, view_num, res, point_size,
                      edge_point_size, mask_ratio_thresh):
    """ours_utils.py:848-882 with save_path=None."""
    sparse = np.zeros((view_num, 3, res, res), F32)
    m0 = np.zeros_like(sparse)
    m2 = np.zeros_like(sparse)
    sf = np.zeros((view_num,), F32)
    for i in range(view_num):
        s, a, b, _, f = get_one_sparse_img(point_pixels[i], colors, point_validation[i], hard_masks[i], res,
                                           point_size, edge_point_size, mask_ratio_thresh)
        sparse[i] = s * a
        m0[i] = a
        m2[i] = b
        sf[i] = f
    return sparse, m0, m2, sf
