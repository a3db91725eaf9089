"""UV-atlas producer: the rasterise-in-UV-space + interpolate half of
models/get3d/extract_texture_map.xatlas_uvmap_w_face_id (/root/reference/models/get3d/extract_texture_map.py:42-64).
The chart parametrisation itself (`xatlas.parametrize`, CPU third-party) stays upstream: pass its `uvs` / `mesh_tex_idx`.
Returns the wire format of demo.py:445-448: uvs, mesh_tex_idx, gb_pos[1,R,R,3], mask[1,R,R,1], per_atlas_pixel_face_id[1,R,R]."""
import torch

from . import _lib
from ._lib import ptr, as_u8, stream, check


def rasterize(pos, tri, resolution):
    """pos [V,Vn,4] f32 clip-space (w = 1), tri [F,3] -> face_idx [V,R,R] i64 (-1 empty), bary [V,R,R,2], depth, mask."""
    L = _lib.lib()
    pos = pos.float().contiguous()
    tri32 = tri.to(torch.int32).contiguous()
    V, Vn = pos.shape[:2]
    R = int(resolution)
    dev = pos.device
    zkey = torch.empty(((L.pdhip_raster_mesh_ws_bytes(V, tri32.shape[0], R) + 7) // 8,), dtype=torch.int64, device=dev)
    hard = torch.empty((V, R, R), dtype=torch.bool, device=dev)
    fidx = torch.empty((V, R, R), dtype=torch.int64, device=dev)
    depth = torch.empty((V, R, R), device=dev)
    check(L.pdhip_raster_mesh_ws(ptr(pos), V, Vn, ptr(tri32), tri32.shape[0], R, ptr(zkey), zkey.numel() * 8, ptr(as_u8(hard)), ptr(fidx),
                                 ptr(depth), stream()), 'pdhip_raster_mesh')
    bary = torch.empty((V, R, R, 2), device=dev)
    check(L.pdhip_raster_barycentrics(ptr(pos), V, Vn, ptr(tri32), R, ptr(fidx), ptr(bary), stream()), 'pdhip_raster_barycentrics')
    return fidx, bary, depth, hard


def interpolate(attr, fidx, bary, tri):
    """nvdiffrast.interpolate(attr[None], rast, tri): attr [Na,C], tri [F,3] -> [V,R,R,C]."""
    L = _lib.lib()
    attr = attr.float().contiguous()
    tri32 = tri.to(torch.int32).contiguous()
    C = attr.shape[1]
    out = torch.empty(tuple(fidx.shape) + (C,), device=attr.device)
    check(L.pdhip_interpolate(ptr(attr), C, ptr(tri32), ptr(fidx.contiguous()), ptr(bary.contiguous()), fidx.numel(), ptr(out),
                              stream()), 'pdhip_interpolate')
    return out


def uvmap_w_face_id(mesh_v, mesh_pos_idx, uvs, mesh_tex_idx, resolution):
    """extract_texture_map.py:48-64 given the parametrisation: rasterise the UV triangles, interpolate world positions."""
    uv_clip = uvs.float()[None] * 2.0 - 1.0
    uv_clip4 = torch.cat((uv_clip, torch.zeros_like(uv_clip[..., 0:1]), torch.ones_like(uv_clip[..., 0:1])), dim=-1).contiguous()
    fidx, bary, _, hard = rasterize(uv_clip4, mesh_tex_idx, resolution)
    gb_pos = interpolate(mesh_v, fidx, bary, mesh_pos_idx)
    return uvs, mesh_tex_idx, gb_pos, hard.unsqueeze(-1), fidx
