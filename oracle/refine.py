"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the reference's optional
`refine_point_validation_by_remove_abnormal_depth` stage --
  * `refine_point_validation`               /root/reference/pointdreamer/ours_utils.py:227-305
  * `detect_abnormal_bright_spots_in_gray_img`  /root/reference/utils/utils_2d.py:584-658 (the part that decides the mask)
  * `get_cam_Ks_RTs_from_locations`         /root/reference/utils/camera_utils.py:940-985
written the slow, literal way (whole-image operations per component, explicit neighbourhood loops) so that it shares no code
path with the product's bounding-box / scipy.ndimage form.

PARITY UNPINNED against OpenCV: cv2 is not installed in this image (the reference imports it), so the four OpenCV calls are
restated from their documented behaviour -- Scharr 3x3 with BORDER_REFLECT_101, convertScaleAbs = saturate(|x|), addWeighted with
cvRound (half to even), connectedComponents 8-connectivity with label 0 = zero pixels, dilate with a 3x3 box ignoring the outside --
and pinned only by the hand-computed cases in tests/test_refine_cpu.py.  Everything else is the reference's own numpy, op for op.
"""
import numpy as np

F32 = np.float32


def get_cam_Ks_RTs_from_locations(cam_locations):
    """camera_utils.py:940-985."""
    loc = np.asarray(cam_locations, dtype=np.float64)
    out = np.zeros((len(loc), 3, 4))
    for i in range(len(loc)):
        eye = loc[i]
        N = -eye / np.linalg.norm(eye)
        up = np.array([0, 0, 1.0]) if (N[0] == 0 and N[2] == 0) else np.array([0, 1.0, 0])
        U = np.cross(N, up); U = U / np.linalg.norm(U)
        V = np.cross(U, N); V = V / np.linalg.norm(V)
        out[i, 0, :3], out[i, 1, :3], out[i, 2, :3] = U, V, N
        out[i, :, 3] = [np.dot(-U, eye), np.dot(-V, eye), np.dot(-N, eye)]
    K = np.array([[560.0, 0, 256], [0, 560, 256], [0, 0, 1]])
    return K, out


def _r101(i, n):
    """BORDER_REFLECT_101 index: ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ..."""
    if i < 0:
        return -i
    if i >= n:
        return 2 * n - 2 - i
    return i


def scharr_abs(u8):
    H, W = u8.shape
    kx = [[-3, 0, 3], [-10, 0, 10], [-3, 0, 3]]
    ax = np.zeros((H, W), np.uint8); ay = np.zeros((H, W), np.uint8)
    a = u8.astype(np.int64)
    for y in range(H):
        for x in range(W):
            gx = gy = 0
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    v = a[_r101(y + dy, H), _r101(x + dx, W)]
                    gx += kx[dy + 1][dx + 1] * v
                    gy += kx[dx + 1][dy + 1] * v
            ax[y, x] = min(abs(gx), 255); ay[y, x] = min(abs(gy), 255)
    return ax, ay


def scharr_abs_fast(u8):
    """Same values by shifted whole-image slices (used for the larger test images; checked against scharr_abs on small ones)."""
    H, W = u8.shape
    idx_y = np.array([_r101(i, H) for i in range(-1, H + 1)]); idx_x = np.array([_r101(i, W) for i in range(-1, W + 1)])
    a = u8.astype(np.int64)[idx_y][:, idx_x]
    s = lambda dy, dx: a[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    gx = 3 * s(-1, 1) + 10 * s(0, 1) + 3 * s(1, 1) - 3 * s(-1, -1) - 10 * s(0, -1) - 3 * s(1, -1)
    gy = 3 * s(1, -1) + 10 * s(1, 0) + 3 * s(1, 1) - 3 * s(-1, -1) - 10 * s(-1, 0) - 3 * s(-1, 1)
    return np.minimum(np.abs(gx), 255).astype(np.uint8), np.minimum(np.abs(gy), 255).astype(np.uint8)


def round_half_even_mean(a, b):
    """cv2.addWeighted(a, .5, b, .5, 0): s = a + b; s/2 rounded half to even."""
    s = a.astype(np.int64) + b.astype(np.int64)
    q, r = s // 2, s % 2
    return np.where(r == 0, q, q + (q % 2)).astype(np.uint8)


def label8(nonzero):
    """Flood fill in raster order, 8-connectivity; 0 = zero pixels.  Returns (num_labels, labels)."""
    H, W = nonzero.shape
    lab = np.zeros((H, W), np.int32)
    n = 0
    for y in range(H):
        for x in range(W):
            if nonzero[y, x] and lab[y, x] == 0:
                n += 1
                lab[y, x] = n
                stack = [(y, x)]
                while stack:
                    cy, cx = stack.pop()
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            yy, xx = cy + dy, cx + dx
                            if 0 <= yy < H and 0 <= xx < W and nonzero[yy, xx] and lab[yy, xx] == 0:
                                lab[yy, xx] = n
                                stack.append((yy, xx))
    return n + 1, lab


def dilate(mask, iterations):
    m = mask.astype(bool)
    H, W = m.shape
    for _ in range(iterations):
        p = np.zeros((H + 2, W + 2), bool)
        p[1:-1, 1:-1] = m
        o = np.zeros((H, W), bool)
        for dy in range(3):
            for dx in range(3):
                o |= p[dy:dy + H, dx:dx + W]
        m = o
    return m


def detect_abnormal_bright_spots(img, foreground_mask, min_for_norm=1.0, max_for_norm=3.0, edge_thresh=50, pixel_num_thresh=200,
                                 area_expand_thresh=5, area_same_color_thres=5, brighter_thresh=6, exhaustive_scharr=False):
    """utils_2d.py:584-658, every component over the whole image like the reference."""
    res = img.shape[0]
    u = (np.asarray(img) - min_for_norm) / (max_for_norm - min_for_norm)
    u = u * 255.0
    u8 = np.clip(u, 0, 255).astype(np.uint8)
    ax, ay = (scharr_abs if exhaustive_scharr else scharr_abs_fast)(u8)
    edges = round_half_even_mean(ax, ay)
    num, labels = label8(edges <= edge_thresh)
    fg = np.asarray(foreground_mask).astype(bool)
    abnormal = np.zeros((res, res), bool)
    for i in range(num):
        area = labels == i
        if area.astype(np.int32).sum() < pixel_num_thresh:
            dil = dilate(area, area_expand_thresh)
            if np.logical_and(dil, ~fg).astype(np.int32).sum() < 1:
                if not area.any():
                    continue
                mean_color = u8[area].astype(np.float64).mean()
                same = np.abs(u8.astype(np.float64) - mean_color) < area_same_color_thres
                final = np.logical_and(dil, same)
                around = np.logical_and(dil, ~final)
                if around.any() and (mean_color - u8[around].mean()) > brighter_thresh:
                    abnormal[final] = True
    return abnormal


def refine_point_validation(cam_RTs, res, hard_masks, point_validation, point_uvs, points, resize_mask, nearest_inpaint):
    """ours_utils.py:227-305.  resize_mask / nearest_inpaint: the oracle's P2b and I0 functions (oracle.project
    .resize_mask_bilinear_nonzero, oracle.inpaint.nearest_inpaint), passed in to keep this module free of imports."""
    point_uvs = np.asarray(point_uvs, F32)
    V, N = point_validation.shape
    pp = point_uvs * F32(res)
    with np.errstate(invalid='ignore'):
        pp = pp.astype(np.int64)
    pp = np.clip(np.stack([pp[:, :, 1], pp[:, :, 0]], -1), 0, res - 1)
    RT = np.asarray(cam_RTs).astype(F32)
    P = np.asarray(points, F32)
    out = np.asarray(point_validation).astype(bool).copy()
    for i in range(V):
        val = np.asarray(point_validation[i]).astype(bool)
        z = ((P[:, 0] * RT[i, 2, 0] + P[:, 1] * RT[i, 2, 1]) + P[:, 2] * RT[i, 2, 2]) + RT[i, 2, 3]
        sparse = np.full((res, res), -100.0, F32)
        for k in np.nonzero(val)[0]:                       # point order: the last point on a pixel wins
            sparse[pp[i, k, 0], pp[i, k, 1]] = z[k]
        non_empty = sparse != -100
        fg = resize_mask(np.asarray(hard_masks[i]).astype(bool), res, res)
        dense = nearest_inpaint(sparse[None], non_empty[None])[0]
        abnormal = detect_abnormal_bright_spots(dense, fg, 0.5, 2.5, 25, 2000, 5, 5, 5)
        out[i][val] = ~abnormal[pp[i, val, 0], pp[i, val, 1]]
    return out
