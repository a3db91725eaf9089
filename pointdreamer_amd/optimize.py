"""optimize_color, same signature and return as the reference
(/root/reference/pointdreamer/ours_utils.py:1583-1785): refine the atlas against the inpainted views."""
import torch

from . import _lib
from ._lib import ptr, as_u8, stream, check
from .camera_utils import stack_params
from .extract_texture_map import rasterize, interpolate
from .ours_utils import crop_params


_DEBUG_COUNTS = None      # set to a list by a tool to receive [masked pixels, active texels, pixels, texels] of each call


def texture_coordinates(cams, vertices, faces, uvs, mesh_tex_idx, uv_centers, uv_scales, padding, inpaint_scale_factors, res):
    """ours_utils.py:1675-1706 (unflipped): uv_map [V,res,res,2], face_idx [V,res,res]."""
    L = _lib.lib()
    dev = vertices.device
    V, Vn = len(cams), vertices.shape[0]
    cp = stack_params(cams)
    verts = vertices.float().contiguous()
    pos = torch.empty((V, Vn, 4), device=dev)
    vuv = torch.empty((V, Vn, 2), device=dev)
    ws = torch.empty((4 * V,), dtype=torch.int32, device=dev)
    check(L.pdhip_project_points(ptr(cp), V, ptr(verts), Vn, None, 0, 0, 0.0, ptr(pos), ptr(vuv), None, None, None, None,
                                 ptr(ws), stream()), 'pdhip_project_points')
    uvc, uvs_, padding, sf = crop_params(V, dev, uv_centers, uv_scales, padding, inpaint_scale_factors)
    check(L.pdhip_rescale_vertices(ptr(pos), V, Vn, ptr(uvc), ptr(uvs_), ptr(sf), float(padding), stream()), 'pdhip_rescale_vertices')
    fidx, bary, _, _ = rasterize(pos, faces, res)
    uv_map = interpolate(uvs, fidx, bary, mesh_tex_idx)
    return uv_map, fidx


def optimize_color(atlas_img, inpainted_imgs, vertices, faces, uvs, mesh_tex_idx, cams, eye_positions, look_ats, up_dirs,
                   uv_centers, uv_scales, padding, inpaint_scale_factors, glctx=None,
                   shrinked_per_view_per_pixel_visibility=None, lr=5e-2, iterations=100, print_every=10, res=1024):
    """atlas_img [3,A,A] (or None = random init, 'scratch') -> (atlas [1,3,A,A], final render [V,3,res,res])."""
    L = _lib.lib()
    dev = vertices.device
    if atlas_img is None:
        atlas = torch.rand((3, 1024, 1024), dtype=torch.float32, device=dev)
    else:
        atlas = atlas_img.detach().float().contiguous().clone()
    A = atlas.shape[2]
    V = len(cams)
    uv_map, fidx = texture_coordinates(cams, vertices, faces, uvs, mesh_tex_idx, uv_centers, uv_scales, padding,
                                       inpaint_scale_factors, res)
    inp = inpainted_imgs.float().contiguous()
    shr = None if shrinked_per_view_per_pixel_visibility is None else as_u8(shrinked_per_view_per_pixel_visibility.contiguous())
    final = torch.empty((V, 3, res, res), device=dev)
    ws = torch.empty((L.pdhip_optimize_color_ws_bytes(V, res, A),), dtype=torch.uint8, device=dev)
    check(L.pdhip_optimize_color(ptr(atlas), A, ptr(uv_map), ptr(fidx), V, res, ptr(inp), inp.shape[-1], ptr(shr, allow_none=True),
                                 float(lr), int(iterations), ptr(final), ptr(ws), stream()), 'pdhip_optimize_color')
    if _DEBUG_COUNTS is not None:                       # tools: (masked pixels, active texels) of the last call -- synchronises
        c = ws[-256:].view(torch.int32)[:2].cpu()
        _DEBUG_COUNTS[:] = [int(c[0]), int(c[1]), V * res * res, A * A]
    return atlas.unsqueeze(0), final
