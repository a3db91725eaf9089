"""optimize_color -- refine the atlas against the inpainted views.  (oracle -- test infrastructure)

Follows /root/reference/pointdreamer/ours_utils.py:1583-1785 (the `use_nvdiffrast_rast` branch): rasterise the mesh per
view at `res` (1024 in the reference) with the crop/rescale transform that includes the per-view inpaint scale factor,
interpolate the per-corner UVs, flip vertically, then `iterations` Adam steps (lr 5e-2, StepLR(15, 0.5)) on the float32
atlas of an L1 loss between bilinear texture lookups (float64) and the views resized to `res`, masked by the foreground
and -- when given -- by the shrunk per-view visibility looked up at the (unflipped) integer texel.
Third-party pieces restated (un-vendored, PARITY UNPINNED): kaolin.render.mesh.texture_mapping(mode='bilinear') ==
grid_sample(2uv-1 with v negated, align_corners=False, padding_mode='border'); nvdiffrast rasterize/interpolate == the
oracle's own rasteriser; torchvision Resize == F.interpolate(bilinear, align_corners=False).
"""
import numpy as np
import torch
import torch.nn.functional as F
from . import project as oproj

F32 = np.float32


def texture_coordinates(cams, vertices, faces, uvs, mesh_tex_idx, uv_centers, uv_scales, padding, inpaint_scale_factors, res):
    """ours_utils.py:1675-1715: returns uv_map[V,res,res,2] f32 and mask[V,res,res] bool, both flipped vertically."""
    vertices = np.asarray(vertices, F32)
    V = len(cams)
    pos = np.zeros((V, vertices.shape[0], 4), F32)
    for i, cam in enumerate(cams):
        pos[i, :, :3] = cam.transform(vertices)
        pos[i, :, 3] = 1.0
    vuv = pos[:, :, :2]
    vuv = (vuv - np.asarray(uv_centers, F32)) / np.asarray(uv_scales, F32)
    vuv = vuv * F32(1 - 2 * padding)
    vuv = vuv * np.asarray(inpaint_scale_factors, F32)[:, None, None]
    vuv = vuv + F32(0.5)
    vuv = np.clip(vuv, F32(0), F32(1))
    pos[:, :, :2] = vuv * F32(2) - F32(1)
    hard, fid, _ = oproj.rasterize(pos, faces, res)
    bary = oproj.raster_barycentrics(pos, faces, fid, res)
    uv_map = oproj.interpolate(uvs, mesh_tex_idx, fid, bary)
    return uv_map[:, ::-1].copy(), hard[:, ::-1].copy()


def texture_mapping_bilinear(texture_coords, atlas):
    """kaolin texture_mapping(mode='bilinear'): texture_coords [B,H,W,2] in [0,1], atlas [B,C,A,A] -> [B,H,W,C]."""
    g = texture_coords * 2.0 - 1.0
    g = torch.stack([g[..., 0], -g[..., 1]], -1)
    out = F.grid_sample(atlas, g, mode='bilinear', align_corners=False, padding_mode='border')
    return out.permute(0, 2, 3, 1)


def optimize_color(atlas_img, inpainted_imgs, uv_map, mask, shrinked_visibility=None, lr=5e-2, iterations=100, res=None):
    """ours_utils.py:1716-1785.  atlas_img [3,A,A] f32, inpainted_imgs [V,3,r,r], uv_map [V,res,res,2], mask [V,res,res].
    Returns atlas [1,3,A,A] f32 and the last rendered images [V,3,res,res] f64."""
    atlas = torch.as_tensor(np.asarray(atlas_img, F32)).unsqueeze(0).clone().requires_grad_()
    tc = torch.as_tensor(np.asarray(uv_map, F32))
    V, res = tc.shape[0], tc.shape[1]
    mask_t = torch.as_tensor(np.asarray(mask)).bool().unsqueeze(-1)                    # [V,res,res,1]
    A = atlas.shape[3]
    tcl = torch.clip((tc * A).long(), 0, A - 1)
    opt = torch.optim.Adam([atlas], lr=lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=15, gamma=0.5)
    target = F.interpolate(torch.as_tensor(np.asarray(inpainted_imgs, F32)), size=(res, res), mode='bilinear', align_corners=False)
    fg = mask_t.permute(0, 3, 1, 2).repeat(1, 3, 1, 1).float()
    target = target * fg
    shr = None
    if shrinked_visibility is not None:
        sv = torch.as_tensor(np.asarray(shrinked_visibility)).bool()
        shr = sv[torch.arange(V)[:, None, None], tcl[..., 1], tcl[..., 0]].unsqueeze(1).float()
        target = target * shr
    images = None
    for it in range(iterations):
        opt.zero_grad()
        images = texture_mapping_bilinear(tc.double(), atlas.repeat(V, 1, 1, 1).double())
        images = torch.clamp(images * mask_t, 0., 1.)
        images = torch.clamp(images, 0., 1.)
        images = images.permute(0, 3, 1, 2)
        images = images * fg
        if shr is not None:
            images = images * shr
        loss = torch.mean(torch.abs(images - target))
        loss.backward()
        opt.step()
        sched.step()
    return atlas.detach(), images.detach()
