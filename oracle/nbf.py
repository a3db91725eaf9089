"""Rows N1, N2, N3: Non-Border-First visibility shrink.  (oracle -- test infrastructure)

Follows /root/reference/utils/utils_2d.py:799-827 (detect_edges_in_gray_by_scharr_torch_batch),
:833-845 (dilate_torch_batch) and /root/reference/pointdreamer/unproject.py:429-475
(get_shrinked_per_view_per_pixel_visibility_torch).

On {0,255} images both reference thresholds (>125 and >126.5 on (|gx|+|gy|)/2) reduce exactly to
"gx != 0 or gy != 0" in integer arithmetic on the 0/1 image (zero padding); the reflect-padded
k x k max-pool is a k x k OR whose out-of-range taps are mirror images of in-range taps, i.e. a
clamped-window OR.  Both identities are checked against the float formulation in tests.
"""
import numpy as np


def scharr_edges_binary(img01):
    """img01[...,H,W] in {0,1} -> bool edges (N1).  Kx=[[-3,0,3],[-10,0,10],[-3,0,3]], Ky=Kx^T, zero pad."""
    a = np.asarray(img01).astype(np.int32)
    p = np.zeros(a.shape[:-2] + (a.shape[-2] + 2, a.shape[-1] + 2), np.int32)
    p[..., 1:-1, 1:-1] = a
    H, W = a.shape[-2:]

    def s(dy, dx):
        return p[..., 1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    gx = 3 * (s(-1, 1) - s(-1, -1)) + 10 * (s(0, 1) - s(0, -1)) + 3 * (s(1, 1) - s(1, -1))
    gy = 3 * (s(1, -1) - s(-1, -1)) + 10 * (s(1, 0) - s(-1, 0)) + 3 * (s(1, 1) - s(-1, 1))
    return (gx != 0) | (gy != 0)


def dilate_binary(mask, k):
    """mask[...,H,W] bool -> k x k OR-dilation with reflect padding (N2) == clamped window."""
    m = np.asarray(mask, bool)
    r = (k - 1) // 2
    H, W = m.shape[-2:]
    out = np.zeros_like(m)
    tmp = np.zeros_like(m)
    for d in range(-r, r + 1):                               # horizontal
        if d >= 0:
            tmp[..., :, :W - d] |= m[..., :, d:]
        else:
            tmp[..., :, -d:] |= m[..., :, :W + d]
    for d in range(-r, r + 1):                               # vertical
        if d >= 0:
            out[..., :H - d, :] |= tmp[..., d:, :]
        else:
            out[..., -d:, :] |= tmp[..., :H + d, :]
    return out


def shrink_visibility(per_pixel_mask, vis_AAV, kernel_sizes):
    """unproject.py:429-475.  per_pixel_mask[A,A] bool, vis_AAV[A,A,V] bool -> [K,V,A,A] bool."""
    vis = np.asarray(vis_AAV, bool).transpose(2, 0, 1)
    if kernel_sizes[0] == 0:
        return vis[None].copy()
    bg_edges = scharr_edges_binary(np.asarray(per_pixel_mask, bool))
    edges = scharr_edges_binary(vis) & ~bg_edges[None]
    out = []
    for k in kernel_sizes:
        border = dilate_binary(edges, k)
        out.append(vis & ~border)
    return np.stack(out, 0)


def shrink_triptychs(per_pixel_mask, vis_AAV, kernel_sizes):
    """unproject.py:459-474 (the `shrink_per_view_edge/{v}.png` debug images) as uint8 [V, A, 3A+20, 3]:
    panel 0 = visibility in grey with the chart-background edges red and the view edges blue, panel 1 = view edge mask,
    panel 2 = border mask of the LAST kernel size (the python loop variable the reference reuses), 10 white columns between
    panels (utils_2d.cat_images: margin 10 on a canvas of ones), rows reversed, x255 truncated."""
    vis = np.asarray(vis_AAV, bool).transpose(2, 0, 1)
    V, A, _ = vis.shape
    bg = scharr_edges_binary(np.asarray(per_pixel_mask, bool))
    edges = scharr_edges_binary(vis) & ~bg[None]
    border = dilate_binary(edges, int(kernel_sizes[-1]))
    out = np.full((V, A, 3 * A + 20, 3), 255, np.uint8)
    p0 = np.repeat(vis[..., None], 3, -1).astype(np.uint8) * 255
    p0[np.broadcast_to(bg[None], vis.shape)] = (255, 0, 0)
    p0[edges] = (0, 0, 255)
    out[:, :, :A] = p0
    out[:, :, A + 10:2 * A + 10] = (edges[..., None] * 255).astype(np.uint8)
    out[:, :, 2 * A + 20:] = (border[..., None] * 255).astype(np.uint8)
    return out[:, ::-1].copy()
