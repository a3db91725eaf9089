"""Host-side image heuristics of the reference's utils/utils_2d.py that the texturing path calls:
`detect_abnormal_bright_spots_in_gray_img` (utils_2d.py:584-707, the blob test behind
`refine_point_validation_by_remove_abnormal_depth`, ours_utils.py:227-305) and `cat_images` (utils_2d.py:94-140).

The reference runs these on the host with OpenCV; cv2 is not a dependency here, so the four OpenCV calls are restated in
numpy / scipy.ndimage with OpenCV's documented conventions:
  * cv2.Scharr(uint8, CV_64F, dx, dy): 3x3 kernel [-3 0 3; -10 0 10; -3 0 3] (and its transpose), BORDER_REFLECT_101;
  * cv2.convertScaleAbs: saturate_cast<uchar>(|x|) (inputs are integers here, so no rounding question);
  * cv2.addWeighted(a, .5, b, .5, 0) on uint8: saturate_cast<uchar>(cvRound(.5 a + .5 b)), cvRound = round-half-to-even;
  * cv2.connectedComponents(img, connectivity=8): label 0 = the zero pixels, 1.. = 8-connected components of the non-zero ones
    (the NUMBERING of the components does not enter the result: the abnormal mask is a union over components);
  * cv2.dilate(mask, ones(3,3), iterations=k): k-fold 3x3 maximum, pixels outside the image never contribute.
Everything after those calls is the reference's own numpy, kept op for op (float32 normalisation, uint8 truncation, float64 means).
Per-component work is restricted to the component's bounding box grown by the dilation radius -- the reference runs every
component over the whole image, which is the same set of pixels.
"""
import numpy as np


def scharr_abs_u8(u8):
    """(|Scharr_x|, |Scharr_y|) of a uint8 image as uint8 (cv2.Scharr -> cv2.convertScaleAbs), BORDER_REFLECT_101."""
    a = np.pad(u8.astype(np.int32), 1, mode='reflect')           # numpy 'reflect' == OpenCV REFLECT_101 (edge pixel not repeated)
    H, W = u8.shape
    s = lambda dy, dx: a[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    gx = 3 * (s(-1, 1) - s(-1, -1)) + 10 * (s(0, 1) - s(0, -1)) + 3 * (s(1, 1) - s(1, -1))
    gy = 3 * (s(1, -1) - s(-1, -1)) + 10 * (s(1, 0) - s(-1, 0)) + 3 * (s(1, 1) - s(-1, 1))
    return np.minimum(np.abs(gx), 255).astype(np.uint8), np.minimum(np.abs(gy), 255).astype(np.uint8)


def add_weighted_half(a, b):
    """cv2.addWeighted(a, 0.5, b, 0.5, 0) on uint8 images: round-half-to-even of the mean."""
    return np.clip(np.rint(0.5 * a.astype(np.float64) + 0.5 * b.astype(np.float64)), 0, 255).astype(np.uint8)


def connected_components8(nonzero):
    """cv2.connectedComponents(connectivity=8): (num_labels, labels int32) with label 0 for the zero pixels."""
    from scipy import ndimage
    labels, n = ndimage.label(nonzero, structure=np.ones((3, 3), dtype=bool))
    return n + 1, labels.astype(np.int32)


def dilate3x3(mask, iterations):
    """cv2.dilate(mask.astype(uint8), ones((3,3)), iterations=k).astype(bool)."""
    from scipy import ndimage
    if iterations <= 0:
        return mask.astype(bool)
    return ndimage.binary_dilation(mask.astype(bool), structure=np.ones((3, 3), dtype=bool), iterations=int(iterations), border_value=0)


def depth_to_u8(img, min_for_norm, max_for_norm):
    """utils_2d.py:602-608: float32 normalisation, clip, uint8 TRUNCATION."""
    u = (np.asarray(img) - min_for_norm) / (max_for_norm - min_for_norm)
    u = u * 255.0
    return np.clip(u, 0, 255).astype(np.uint8)


def detect_abnormal_bright_spots_in_gray_img(img, foreground_mask, save_path=None, min_for_norm=1.0, max_for_norm=3.0, edge_thresh=50,
                                             pixel_num_thresh=200, area_expand_thresh=5, area_same_color_thres=5, brighter_thresh=6,
                                             _details=None):
    """utils_2d.py:584-707.  img [res,res] (the nearest-filled depth map), foreground_mask [res,res] bool ->
    abnormal_mask [res,res] bool: small regions bounded by Scharr edges, entirely inside the foreground, brighter than their
    surroundings.  save_path: the reference's four-panel picture (original | edges | regions | cleaned depth), flipped
    upside down; the region colours are random in the reference too."""
    res = img.shape[0]
    uint8_img = depth_to_u8(img, min_for_norm, max_for_norm)
    ax, ay = scharr_abs_u8(uint8_img)
    edges = add_weighted_half(ax, ay)
    num_labels, labels = connected_components8(edges <= edge_thresh)          # edges_binary = 255 where edges <= thresh
    foreground_mask = np.asarray(foreground_mask).astype(bool)
    abnormal_mask = np.zeros((res, res), dtype=bool)
    from scipy import ndimage
    boxes = ndimage.find_objects(labels) if num_labels > 1 else []
    counts = np.bincount(labels.ravel(), minlength=num_labels)
    abnormal_labels = []
    r = max(int(area_expand_thresh), 0)
    for i in range(num_labels):
        if counts[i] >= pixel_num_thresh:                                      # a bright spot is smaller than the threshold
            continue
        if i == 0:                                                             # label 0 = the edge pixels themselves: the reference
            if counts[0] == 0:                                                 # tests them as one region too (range(num_labels))
                continue
            ys, xs = np.nonzero(labels == 0)
            box = (slice(ys.min(), ys.max() + 1), slice(xs.min(), xs.max() + 1))
        else:
            box = boxes[i - 1]
        y0, y1 = max(box[0].start - r, 0), min(box[0].stop + r, res)
        x0, x1 = max(box[1].start - r, 0), min(box[1].stop + r, img.shape[1])
        win = (slice(y0, y1), slice(x0, x1))
        label_area = labels[win] == i
        dil = dilate3x3(label_area, area_expand_thresh)                        # (the window holds the whole dilation)
        if np.logical_and(dil, ~foreground_mask[win]).any():                   # only spots inside the foreground
            continue
        u = uint8_img[win]
        mean_color = u[label_area].astype(np.float64).mean()
        same = np.abs(u.astype(np.float64) - mean_color) < area_same_color_thres
        final = np.logical_and(dil, same)
        around = np.logical_and(dil, ~final)
        if not around.any():
            continue                                                           # (mean of nothing is nan in the reference: the test below fails)
        if (mean_color - u[around].mean()) > brighter_thresh:                  # brighter than the pixels around it
            abnormal_mask[win] |= final
            abnormal_labels.append((i, win, final))
    if _details is not None:
        _details.update(uint8_img=uint8_img, edges=edges, labels=labels, num_labels=num_labels)
    if save_path is not None:
        _save_panels(save_path, uint8_img, edges, edge_thresh, labels, abnormal_labels, abnormal_mask, foreground_mask)
    return abnormal_mask


def _save_panels(path, uint8_img, edges, edge_thresh, labels, abnormal_labels, abnormal_mask, foreground_mask):
    """utils_2d.py:660-706: original | edges in red | abnormal regions in random colours | depth with the spots filled from their
    nearest normal pixel, background zeroed, flipped upside down."""
    import torch
    from . import io_utils, ours_utils as ou
    src = np.repeat(uint8_img[..., None], 3, axis=2)                           # cv2.cvtColor(GRAY2BGR)
    color_img = src.copy()
    colors = np.random.randint(0, 255, size=(int(labels.max()) + 1, 3), dtype=np.uint8)
    for i, win, final in abnormal_labels:
        c = color_img[win]
        c[labels[win] == i] = 0
        c[final] = colors[i]
    dev = torch.device('cuda')
    dense = ou.nearest_fill(torch.from_numpy(uint8_img.astype(np.float32))[None, None].to(dev),
                            torch.from_numpy(~abnormal_mask)[None].to(dev))[0, 0].cpu().numpy()
    dense[~foreground_mask] = 0
    depth_map_img = np.repeat(dense[None], 3, axis=0).astype(np.float64)
    with_edges = src.copy()
    with_edges[edges > edge_thresh] = (0, 0, 255)
    fg = foreground_mask[..., None]
    cat = cat_images((src * fg).transpose(2, 0, 1), (with_edges * fg).transpose(2, 0, 1)) / 255.0
    cat = cat_images(cat, (color_img * fg).transpose(2, 0, 1) / 255.0)
    cat = cat_images(cat, depth_map_img / 255.0)
    io_utils.save_CHW_RGB_img(np.ascontiguousarray(np.flip(cat, 1)), path)


def cat_images(img1, img2, margin=10, horizon=True):
    """utils_2d.py:94-140: [C,H,W] images side by side (or stacked) on a white canvas with a margin; img2 is resized to img1's
    height (width) first -- bilinear like torchvision's Resize."""
    img1, img2 = np.asarray(img1, dtype=np.float64), np.asarray(img2, dtype=np.float64)
    _, h1, w1 = img1.shape
    _, h2, w2 = img2.shape
    tgt = (h1, int(w2 * h1 / h2)) if horizon else (int(h2 * w1 / w2), w1)
    if tgt != (h2, w2):
        import torch
        img2 = torch.nn.functional.interpolate(torch.from_numpy(img2)[None], size=tgt, mode='bilinear', align_corners=False,
                                               antialias=True)[0].numpy()
        _, h2, w2 = img2.shape
    if horizon:
        out = np.ones((img1.shape[0], h1, w1 + margin + w2))
        out[:, :h1, :w1] = img1
        out[:, :h2, w1 + margin:] = img2
    else:
        out = np.ones((img1.shape[0], h1 + margin + h2, w1))
        out[:, :h1, :w1] = img1
        out[:, h1 + margin:, :w2] = img2
    return out
