"""Host-side mirror of the reference's DDNM inpainter (rows I1, D1, U1):
models/DDNM/ddnm_inpainting.py:15-44 (`Inpainter(device).inpaint(masked_imgs, masks)`),
guided_diffusion/diffusion.py:435-570 (`get_model`, `simplified_ddnm_inpainting`) and
guided_diffusion/script_util.py:130-185 (`create_model`) with configs/imagenet_256.yml.

The network and the 100-step sampling loop run inside libpdhip.so (pdhip_unet_*, pdhip_ddnm_sample);
this module only resolves the configuration, hands the state dict over by the reference's key names and
keeps the reference's call signatures.  Views are batched through the UNet (the reference's serial
batch-1 loop is a convenience, not a semantic requirement -- SURVEY 7.6).
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import ptr, stream, check

_reg = _lib.register
vp, i32, i64, f32, f64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_uint64
_reg('pdhip_unet_create', C.c_int, [i32, i32, i32, C.POINTER(C.c_int), i32, C.POINTER(C.c_int), i32, i32, i32, i32,
                                    C.POINTER(vp)])
_reg('pdhip_unet_destroy', None, [vp])
_reg('pdhip_unet_arena_bytes', C.c_longlong, [vp])
_reg('pdhip_unet_num_tensors', C.c_int, [vp])
_reg('pdhip_unet_load_tensor', C.c_int, [vp, C.c_char_p, vp, i32, C.POINTER(C.c_int64), i32, vp])
_reg('pdhip_unet_missing_tensors', C.c_int, [vp, C.c_char_p, i32])
_reg('pdhip_unet_forward', C.c_int, [vp, vp, vp, i32, vp, vp])
_reg('pdhip_unet_profile', C.c_int, [vp, i32])
_reg('pdhip_unet_profile_read', C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)])
_reg('pdhip_unet_profile_read_attention', C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)])
_reg('pdhip_ddnm_schedule', C.c_int, [vp, vp, vp, vp, vp])
_reg('pdhip_ddnm_prepare', C.c_int, [vp, vp, vp, i32, i32, vp])
_reg('pdhip_ddnm_step', C.c_int, [vp, vp, i32, vp, vp, vp, u64, i32, i32, i32, vp])
_reg('pdhip_ddnm_sample', C.c_int, [vp, vp, vp, i32, vp, vp, u64, i32, vp, vp])
_reg('pdhip_ddnm_sample_keyed', C.c_int, [vp, vp, vp, i32, vp, vp, u64, u64, i32, vp, vp])
_reg('pdhip_debug_set_conv_bk', C.c_int, [i32])
_reg('pdhip_debug_set_conv_stages', C.c_int, [i32])
_reg('pdhip_debug_set_conv_tile', C.c_int, [i32])
_reg('pdhip_debug_set_conv_halo_strips', C.c_int, [i32])
_reg('pdhip_debug_set_conv_splitk', C.c_int, [vp, C.c_longlong, i32])
_reg('pdhip_pack_conv_weight_f16', C.c_int, [vp, i32, i32, i32, vp, vp])
_reg('pdhip_unet_head_ws_floats', C.c_size_t, [i32, i32, i32, i32, i32])
_reg('pdhip_unet_head_f32', C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, C.c_longlong, vp])
_reg('pdhip_conv2d_nhwc_f16', C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp])
_reg('pdhip_groupnorm_nhwc_f16', C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, C.c_longlong, vp])
_reg('pdhip_gn_silu_conv3x3_nhwc_f16', C.c_int, [vp, vp, vp, vp, C.c_longlong, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, C.c_longlong, vp])
_reg('pdhip_debug_set_fuse_gn', C.c_int, [i32])
_reg('pdhip_gn_silu_skip1x1_nhwc_f16', C.c_int, [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp])
_reg('pdhip_debug_set_fuse_skip', C.c_int, [i32, i32])
_reg('pdhip_debug_set_gn_skip_variant', C.c_int, [i32])
_reg('pdhip_debug_set_fold_resample', C.c_int, [i32])
_reg('pdhip_debug_set_fold_finalize', C.c_int, [i32])
_reg('pdhip_debug_set_fold_finalize_chunks', C.c_int, [i32])
_reg('pdhip_debug_set_fold_skip', C.c_int, [i32])
_reg('pdhip_debug_set_conv_sk', C.c_int, [i32, i32, i32])
_reg('pdhip_debug_set_conv_sk_stages', C.c_int, [i32])
_reg('pdhip_debug_set_conv_sk_kgroups', C.c_int, [i32])
_reg('pdhip_debug_set_conv_sk_order', C.c_int, [i32])
_reg('pdhip_debug_set_attn', C.c_int, [i32, i32, i32])
_reg('pdhip_debug_set_conv_rr', C.c_int, [i32, i32, i32])
_reg('pdhip_debug_set_rr_gn', C.c_int, [i32])
_reg('pdhip_debug_set_conv_ht', C.c_int, [i32, i32])
_reg('pdhip_conv_ht_plan', C.c_int, [i32, i32, i32, i32, i32, i32, C.c_longlong, C.POINTER(C.c_int), C.POINTER(C.c_int)])
_reg('pdhip_conv_ht_f16', C.c_int, [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, C.c_longlong, vp, C.POINTER(C.c_int), vp])
_reg('pdhip_conv_rr_weight_halfs', C.c_longlong, [i32, i32, i32, i32])
_reg('pdhip_conv_rr_pack_f16', C.c_int, [vp, i32, i32, i32, i32, vp, vp])
_reg('pdhip_gn_octet_partials_f16', C.c_int, [vp, i32, i32, i32, i32, vp, vp])
_reg('pdhip_gn_apply_parts_f16', C.c_int, [vp, vp, i32, i32, vp, i32, vp, i32, vp, vp, vp, C.c_longlong, i32, i32, i32, i32, vp, vp])
_reg('pdhip_conv_rr_f16', C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp, C.c_longlong, vp, i32, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp,
                                   i32, i32, i32, i32, vp, C.c_longlong, vp, C.POINTER(C.c_int), vp])
_reg('pdhip_debug_conv3x3_apply', C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp])
_reg('pdhip_attention_f16', C.c_int, [vp, vp, i32, i32, i32, i32, vp, vp])
_reg('pdhip_philox_normal', C.c_int, [vp, C.c_longlong, u64, u64, vp])
_reg('pdhip_debug_set_gn_iters', C.c_int, [i32])
_reg('pdhip_bench_copy16', C.c_int, [vp, vp, C.c_longlong, i32, i32, vp])

# models/DDNM/configs/imagenet_256.yml (model + diffusion + time_travel sections); values must be reproduced
IMAGENET_256 = dict(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8",
                    num_head_channels=64, learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True,
                    use_fp16=True, channel_mult="")
DEFAULT_CKPT = 'models/DDNM/256x256_diffusion_uncond.pt'


def default_channel_mult(image_size):
    return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[image_size]


class UNetModel:
    """The guided-diffusion UNet as an opaque engine handle.  forward(x[N,3,S,S] f32, t[N]) -> [N,out_ch,S,S] f32."""

    def __init__(self, image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8",
                 num_head_channels=64, learn_sigma=True, channel_mult="", max_batch=8, device='cuda', **unused):
        if not unused.get('use_scale_shift_norm', True) or not unused.get('resblock_updown', True):
            raise NotImplementedError("only the scale-shift-norm / resblock-updown variant (imagenet_256.yml) is built")
        L = _lib.lib()
        cm = tuple(int(c) for c in channel_mult.split(",")) if channel_mult else default_channel_mult(image_size)
        if any(int(c) != c for c in cm):
            raise NotImplementedError("fractional channel multipliers (image_size 512) are not built")
        ads = tuple(image_size // int(r) for r in attention_resolutions.split(","))
        self.image_size, self.out_channels, self.max_batch = image_size, 6 if learn_sigma else 3, max_batch
        self.device = _lib.resolve_device(device)
        cm_arr = (C.c_int * len(cm))(*[int(c) for c in cm])
        ad_arr = (C.c_int * len(ads))(*ads)
        h = vp()
        with torch.cuda.device(self.device):
            check(L.pdhip_unet_create(image_size, num_channels, num_res_blocks, cm_arr, len(cm), ad_arr, len(ads),
                                      num_head_channels, self.out_channels, max_batch, C.byref(h)), 'pdhip_unet_create')
        self._h = h
        self._L = L

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._L.pdhip_unet_destroy(h)

    @property
    def arena_bytes(self):
        return int(self._L.pdhip_unet_arena_bytes(self._h))

    def load_state_dict(self, state_dict, strict=True):
        """Same contract as nn.Module.load_state_dict for the reference's key names (diffusion.py:453)."""
        L = self._L
        n = 0
        with torch.cuda.device(self.device):
            for name, t in state_dict.items():
                if t.dtype not in (torch.float32, torch.float16):
                    t = t.float()
                t = t.to(self.device).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                rc = L.pdhip_unet_load_tensor(self._h, name.encode(), ptr(t), 1 if t.dtype == torch.float16 else 0, shape,
                                              t.dim(), stream())
                if rc == -5 and not strict:            # PDHIP_E_UNKNOWN_NAME: nn.Module.load_state_dict(strict=False) ignores unexpected keys
                    continue
                check(rc, f'pdhip_unet_load_tensor({name})')
                n += 1
            torch.cuda.synchronize(self.device)
        buf = C.create_string_buffer(4096)
        missing = L.pdhip_unet_missing_tensors(self._h, buf, 4096)
        if strict and missing:
            raise _lib.PdhipError(f"load_state_dict: {missing} tensors missing, e.g. {buf.value.decode().split()[:4]}")
        return n

    def forward(self, x, timesteps):
        x = x.float().contiguous()
        t = timesteps.float().to(x.device).contiguous()
        N = x.shape[0]
        if x.device != self.device:
            raise _lib.PdhipError(f"UNetModel lives on {self.device}, input is on {x.device}")
        out = torch.empty((N, self.out_channels, self.image_size, self.image_size), device=x.device)
        with torch.cuda.device(self.device):
            check(self._L.pdhip_unet_forward(self._h, ptr(x), ptr(t), N, ptr(out), stream()), 'pdhip_unet_forward')
        return out

    __call__ = forward

    def profile(self, enable):
        """enable: False / 0 off, True / 1 events around every forward's launches, k > 1 around every k-th forward's."""
        check(self._L.pdhip_unet_profile(self._h, int(enable)), 'pdhip_unet_profile')

    def profile_read(self, attention=False):
        ms, fl, n = C.c_double(), C.c_double(), C.c_longlong()
        fn = self._L.pdhip_unet_profile_read_attention if attention else self._L.pdhip_unet_profile_read
        check(fn(self._h, C.byref(ms), C.byref(fl), C.byref(n)), 'pdhip_unet_profile_read')
        return ms.value, fl.value, n.value


def ddnm_schedule():
    """(at[100], at_next[100], t[100], t_next[100], coefs[100,6]) exactly as the engine uses them (host arrays)."""
    import numpy as np
    L = _lib.lib()
    at = np.zeros(100, np.float32)
    an = np.zeros(100, np.float32)
    t = np.zeros(100, np.int32)
    tn = np.zeros(100, np.int32)
    co = np.zeros((100, 6), np.float32)
    check(L.pdhip_ddnm_schedule(at.ctypes.data, an.ctypes.data, t.ctypes.data, tn.ctypes.data, co.ctypes.data),
          'pdhip_ddnm_schedule')
    return at, an, t, tn, co


def random_state_dict(model_kwargs, seed=0, device='cpu'):
    """Seeded stand-in weights under the reference's key names (the OpenAI checkpoint cannot be downloaded
    offline): fan-in scaled normal conv/linear weights, near-identity GroupNorm affine, and the reference's
    zero-initialised layers re-randomised.  Used by bench.py / smoke only -- tests use the oracle's generator."""
    image_size = model_kwargs.get('image_size', 256)
    mc = model_kwargs.get('num_channels', 256)
    nres = model_kwargs.get('num_res_blocks', 2)
    cm = model_kwargs.get('channel_mult') or ""
    cm = tuple(int(c) for c in cm.split(",")) if cm else default_channel_mult(image_size)
    ads = tuple(image_size // int(r) for r in model_kwargs.get('attention_resolutions', "32,16,8").split(","))
    out_ch = 6 if model_kwargs.get('learn_sigma', True) else 3
    ted = mc * 4
    g = torch.Generator(device='cpu').manual_seed(seed)
    sd = {}

    def lin(name, o, i):
        sd[name + '.weight'] = torch.randn((o, i), generator=g) / math.sqrt(i)
        sd[name + '.bias'] = torch.randn((o,), generator=g) * 0.05

    def conv(name, o, i, k, one_d=False):
        shape = (o, i, k) if one_d else (o, i, k, k)
        sd[name + '.weight'] = torch.randn(shape, generator=g) / math.sqrt(i * k * (1 if one_d else k))
        sd[name + '.bias'] = torch.randn((o,), generator=g) * 0.05

    def norm(name, c):
        sd[name + '.weight'] = 1.0 + torch.randn((c,), generator=g) * 0.05
        sd[name + '.bias'] = torch.randn((c,), generator=g) * 0.05

    def res(name, cin, cout):
        norm(name + '.in_layers.0', cin); conv(name + '.in_layers.2', cout, cin, 3)
        lin(name + '.emb_layers.1', 2 * cout, ted)
        norm(name + '.out_layers.0', cout); conv(name + '.out_layers.3', cout, cout, 3)
        if cin != cout:
            conv(name + '.skip_connection', cout, cin, 1)

    def att(name, c):
        norm(name + '.norm', c); conv(name + '.qkv', 3 * c, c, 1, True); conv(name + '.proj_out', c, c, 1, True)
    lin('time_embed.0', ted, mc); lin('time_embed.2', ted, ted)
    ch = int(cm[0] * mc)
    conv('input_blocks.0.0', ch, 3, 3)
    chans = [ch]
    ds, n = 1, 1
    for level, mult in enumerate(cm):
        for _ in range(nres):
            res(f'input_blocks.{n}.0', ch, int(mult * mc)); ch = int(mult * mc)
            if ds in ads:
                att(f'input_blocks.{n}.1', ch)
            chans.append(ch); n += 1
        if level != len(cm) - 1:
            res(f'input_blocks.{n}.0', ch, ch); chans.append(ch); ds *= 2; n += 1
    res('middle_block.0', ch, ch); att('middle_block.1', ch); res('middle_block.2', ch, ch)
    n = 0
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(nres + 1):
            ich = chans.pop()
            res(f'output_blocks.{n}.0', ch + ich, int(mc * mult)); ch = int(mc * mult)
            k = 1
            if ds in ads:
                att(f'output_blocks.{n}.{k}', ch); k += 1
            if level and i == nres:
                res(f'output_blocks.{n}.{k}', ch, ch); ds //= 2
            n += 1
    norm('out.0', ch); conv('out.2', out_ch, ch, 3)
    return {k: v.to(device) for k, v in sd.items()}


class Inpainter:
    """Drop-in for models/DDNM/ddnm_inpainting.Inpainter: Inpainter(device).inpaint(masked_imgs, masks).

    ckpt_path: a state dict saved with the reference's key names (the OpenAI 256x256_diffusion_uncond.pt loads
    as is).  If the file is absent (no network here) and allow_random_weights is set, seeded random weights of the
    same architecture are used (bench / smoke); otherwise this raises instead of downloading."""

    def __init__(self, device, ckpt_path=DEFAULT_CKPT, model_kwargs=None, max_batch=8, allow_random_weights=False,
                 seed=1234, state_dict=None):
        self.device = _lib.resolve_device(device)
        kw = dict(IMAGENET_256)
        kw.update(model_kwargs or {})
        self.model = UNetModel(max_batch=max_batch, device=self.device, **kw)
        if state_dict is None:
            if ckpt_path and os.path.exists(ckpt_path):
                state_dict = torch.load(ckpt_path, map_location='cpu')
            elif allow_random_weights:
                state_dict = random_state_dict(kw, seed=0)
            else:
                raise FileNotFoundError(f"{ckpt_path} not found (the reference downloads it; there is no network here). "
                                        "Pass allow_random_weights=True for a random-weight run.")
        self.model.load_state_dict(state_dict, strict=True)
        self.seed = seed
        self.n_steps = 100
        self._images = 0           # images inpainted so far = the noise key of the next one
        self.max_batch = max_batch

    def inpaint_views(self, masked_imgs, masks, x_T=None, eps_tape=None, n_steps=None, first_key=None, advance=None):
        """masked_imgs [V,3,r,r] in [0,1], masks [V,r,r] (1 = keep) -> [V,3,r,r].  All V views go through the
        100-step sampler together (chunks of max_batch).

        Noise: view k of this call draws the Philox stream (seed, key = first_key + k); by default first_key is the number of
        images this Inpainter has processed, so successive calls get fresh noise and the result does not depend on how the
        views are chunked or batched.  Sharded callers (dist.py) pass the key of their first view and `advance` = the number
        of views of the whole shape so that every rank stays in step."""
        L = _lib.lib()
        masked_imgs = masked_imgs.float().contiguous()
        masks = masks.float().contiguous()
        V = masked_imgs.shape[0]
        out = torch.empty_like(masked_imgs)
        steps = int(n_steps or self.n_steps)
        key0 = self._images if first_key is None else int(first_key)
        with torch.cuda.device(self.device):
            for s in range(0, V, self.max_batch):
                e = min(V, s + self.max_batch)
                xt = None if x_T is None else x_T[s:e].float().contiguous()
                tape = None if eps_tape is None else eps_tape[:, s:e].float().contiguous()
                check(L.pdhip_ddnm_sample_keyed(self.model._h, ptr(masked_imgs[s:e]), ptr(masks[s:e]), e - s, ptr(xt, allow_none=True),
                                                ptr(tape, allow_none=True), self.seed, key0 + s, steps, ptr(out[s:e]),
                                                stream()), 'pdhip_ddnm_sample')
        self._images = (key0 + V) if advance is None else (self._images + int(advance))
        return out

    def inpaint(self, masked_imgs, masks):
        """Reference signature (ddnm_inpainting.py:29-44): masked_imgs [1,H,W,3], masks [1,H,W,3] -> [1,3,H,W]."""
        imgs = masked_imgs.permute(0, 3, 1, 2).contiguous()
        return self.inpaint_views(imgs, masks[:, :, :, 0].contiguous())
