"""Row P3b: hidden-point removal (Katz et al.), the reference's Open3D call at ours_utils.py:204-225.

The reference copies the cloud to the host and runs qhull once per view; here all views are answered on the device by
`pdhip_hidden_point_removal` (csrc/hpr.hip): spherical flip in float64, then one certified GJK containment query per
point -- against a coarse hull on the matrix cores first, then against the Morton-sorted points outside it."""
import numpy as np
import torch

from . import _lib
from ._lib import ptr, as_u8, stream, check


_EYES = {}


def hidden_point_removal(points, eye_positions, radius, already_valid=None, return_stats=False):
    """points [N,3] (GPU), eye_positions [V,3] (numpy / list, as create_cameras returns them) -> [V,N] bool.
    already_valid [V,N] bool (optional): points another test accepted; they are not queried and the result is the OR.
    return_stats: also return dict(exact_fallback=, unresolved=, fallback_rounds=, distance_f64=) of the certified-verdict machinery
    (synchronises; `unresolved` > 0 means exactly degenerate input whose verdict -- hidden -- is a convention)."""
    L = _lib.lib()
    pts = points.detach().float().contiguous()
    if not pts.is_cuda:
        raise _lib.PdhipError("hidden_point_removal needs a GPU tensor; there is no CPU path")
    eyes_h = np.ascontiguousarray(np.asarray(eye_positions, np.float64).reshape(-1, 3))
    key = (eyes_h.tobytes(), str(pts.device))
    eyes = _EYES.get(key)                                             # (the camera set is fixed for a run: one upload, not one per shape)
    if eyes is None:
        if len(_EYES) > 64:
            _EYES.clear()
        eyes = _EYES[key] = torch.from_numpy(eyes_h).to(pts.device).contiguous()
        _lib._settle(eyes)
    V, N = eyes.shape[0], pts.shape[0]
    vis = torch.empty((V, N), dtype=torch.bool, device=pts.device)   # (every verdict is written by the kernels)
    ws = torch.empty((L.pdhip_hpr_ws_bytes(V, N),), dtype=torch.uint8, device=pts.device)
    skip = None if already_valid is None else as_u8(already_valid.contiguous())
    check(L.pdhip_hidden_point_removal(ptr(pts), N, ptr(eyes), V, float(radius), ptr(skip, allow_none=True), ptr(as_u8(vis)), ptr(ws),
                                       stream()),
          'pdhip_hidden_point_removal')
    if return_stats:
        import ctypes as C
        out = (C.c_longlong * 4)()
        check(L.pdhip_hpr_read_counters(ptr(ws), V, out, stream()), 'pdhip_hpr_read_counters')
        return vis, dict(exact_fallback=int(out[0]), unresolved=int(out[1]), fallback_rounds=int(out[2]), distance_f64=int(out[3]))
    return vis
