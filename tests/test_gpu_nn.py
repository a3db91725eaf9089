"""GPU parity tests for rows U1 (UNet) and D1 (DDNM): HIP kernels through the C ABI vs a plain torch fp32
reference of the same op, vs the oracle (oracle/unet.py, oracle/ddnm.py) and vs golden vectors produced by the
imported reference (tools/gen_golden_nn.py).

Tolerances: the engine computes in f16 with f32 accumulation like the reference's fp16 torso, the
oracle is fp32 => per-forward relative L-inf <= 2e-2 and relative L2 <= 5e-3 on random weights; single kernels
are compared against an fp32 reference fed the same f16-rounded inputs (<= 2e-3 of the output scale);
the DDNM update with injected noise is f32 elementwise => 1e-5."""
import ctypes as C
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, note_measured, U1_FP32_LINF, U1_FP32_L2, U1_SMALL_FP32_LINF, U1_SMALL_FP32_L2
from oracle import unet as ounet, ddnm as oddnm

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope="module")
def nn():
    assert torch.cuda.is_available()
    from pointdreamer_amd import _lib
    import pointdreamer_amd.ddnm_inpainting as di
    return dict(L=_lib.lib(), lib=_lib, di=di)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def hip_conv(nn, x_nchw, w, b, res_nchw=None):
    """x [N,Cin,H,W] f32 (already f16-representable), w [Cout,Cin,k,k]; returns [N,Cout,H,W] f32."""
    L = nn['L']
    N, Cin, H, W = x_nchw.shape
    Cout, taps = w.shape[0], w.shape[2] * w.shape[3]
    pad = ((Cout + 127) // 128) * 128
    x = x_nchw.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    wp = torch.zeros((pad, taps * Cin), dtype=torch.float16, device=DEV)
    wd = w.contiguous().float().to(DEV)
    assert L.pdhip_pack_conv_weight_f16(_ptr(wd), Cout, Cin, taps, _ptr(wp), _stream()) == 0
    bd = b.float().to(DEV)
    y = torch.empty((N, H, W, Cout), dtype=torch.float16, device=DEV)
    zp = torch.zeros((128,), dtype=torch.float16, device=DEV)
    r = None if res_nchw is None else res_nchw.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    rc = L.pdhip_conv2d_nhwc_f16(_ptr(x), _ptr(wp), _ptr(bd), _ptr(r) if r is not None else None, _ptr(y), N, H, W, Cin, Cout,
                                 pad, taps, _ptr(zp), _stream())
    assert rc == 0, L.pdhip_last_error()
    torch.cuda.synchronize()
    return y.float().cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,res", [
    (1, 16, 16, 32, 64, 3, False),      # small
    (1, 8, 8, 64, 96, 3, True),         # M = 64 < tile, Cout not a multiple of 128, fused residual
    (2, 32, 32, 256, 256, 3, True),     # the dominant layer shape at reduced size
    (3, 8, 8, 160, 128, 1, False),      # 1x1 (skip connection / qkv / proj), M = 192 (ragged tile)
    (1, 64, 64, 96, 32, 3, False),      # wide image, narrow output
    (1, 16, 16, 1024, 512, 1, True),    # deep K
])
def test_conv_igemm_vs_torch_fp32(nn, N, H, W, Cin, Cout, k, res):
    g = torch.Generator().manual_seed(N * 1000 + Cin + Cout + k)
    x = (torch.randn((N, Cin, H, W), generator=g)).half().float()
    # asymmetric, transpose-detecting weights
    w = (torch.randn((Cout, Cin, k, k), generator=g) / math.sqrt(Cin * k * k)).half().float()
    b = (torch.randn((Cout,), generator=g) * 0.1).half().float()
    r = (torch.randn((N, Cout, H, W), generator=g)).half().float() if res else None
    ref = F.conv2d(x, w, b, padding=k // 2)
    ref = ref.half().float()
    if res:
        ref = (ref + r).half().float()
    out = hip_conv(nn, x, w, b, r)
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() <= 2e-3 * scale + 1e-3
    # the zero-padding border must be exact zeros' contribution: compare a border pixel separately
    assert torch.allclose(out[:, :, 0, 0], ref[:, :, 0, 0], atol=2e-3 * scale + 1e-3)


@pytest.mark.parametrize("N,H,W,Cc,film,silu,resample", [
    (2, 16, 16, 64, False, True, 0), (1, 8, 8, 96, True, True, 0), (2, 16, 16, 256, False, True, 1),
    (1, 8, 8, 512, False, True, 2), (3, 4, 4, 32, True, False, 0), (1, 32, 32, 2048, False, False, 0),
])
def test_groupnorm_silu_film_resample_vs_torch(nn, N, H, W, Cc, film, silu, resample):
    L = nn['L']
    g = torch.Generator().manual_seed(Cc + H)
    x = (torch.randn((N, Cc, H, W), generator=g) * 1.7 + 0.3).half().float()
    gamma = 1 + 0.1 * torch.randn((Cc,), generator=g)
    beta = 0.1 * torch.randn((Cc,), generator=g)
    fl = 0.3 * torch.randn((N, 2 * Cc), generator=g) if film else None
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    if film:
        ref = ref * (1 + fl[:, :Cc, None, None]) + fl[:, Cc:, None, None]
    if silu:
        ref = F.silu(ref)
    if resample == 1:
        ref = F.avg_pool2d(ref, 2)
    elif resample == 2:
        ref = F.interpolate(ref, scale_factor=2, mode='nearest')
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    Ho, Wo = ref.shape[2:]
    y = torch.empty((N, Ho, Wo, Cc), dtype=torch.float16, device=DEV)
    stats = torch.empty((N * 64,), device=DEV)
    ws = torch.empty((N * 64 * ((H * W + 255) // 256),), device=DEV)
    fd = fl.to(DEV) if film else None
    gd, bd = gamma.to(DEV), beta.to(DEV)                   # keep the device copies alive across the async launch
    rc = L.pdhip_groupnorm_nhwc_f16(_ptr(xd), _ptr(gd), _ptr(bd), _ptr(fd) if film else None, N, H, W, Cc,
                                    1 if silu else 0, resample, _ptr(y), _ptr(stats), _ptr(ws), ws.numel(), _stream())
    assert rc == 0, L.pdhip_last_error()
    out = y.float().cpu().permute(0, 3, 1, 2)
    assert (out - ref).abs().max().item() <= 6e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("N,H,W,Cc,Cout", [(2, 32, 32, 256, 6), (1, 24, 40, 64, 3), (1, 16, 64, 128, 6), (3, 8, 32, 32, 6)])
def test_output_head_f32_equivalent_vs_torch(nn, N, H, W, Cc, Cout):
    """GN -> SiLU -> conv3x3 head (unet.py:613-617, f32 in the reference): the fused kernel's f16 hi/lo split must be
    f32-accurate (1e-5 of the output scale), including tiles that straddle the image border (H, W not multiples of 8/32)."""
    L = nn['L']
    g = torch.Generator().manual_seed(Cc + H + Cout)
    x = (torch.randn((N, Cc, H, W), generator=g) * 1.3 + 0.2).half().float()
    gamma = 1 + 0.2 * torch.randn((Cc,), generator=g)
    beta = 0.2 * torch.randn((Cc,), generator=g)
    w = torch.randn((Cout, Cc, 3, 3), generator=g) / math.sqrt(9 * Cc)
    b = 0.1 * torch.randn((Cout,), generator=g)
    ref = F.conv2d(F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), eps=1e-5)), w.double(), b.double(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    gd, bd, wd, cd = gamma.to(DEV), beta.to(DEV), w.contiguous().to(DEV), b.to(DEV)
    y = torch.empty((N, Cout, H, W), device=DEV)
    ws = torch.empty((L.pdhip_unet_head_ws_floats(N, H, W, Cc, Cout),), device=DEV)
    rc = L.pdhip_unet_head_f32(_ptr(xd), _ptr(gd), _ptr(bd), _ptr(wd), _ptr(cd), N, H, W, Cc, Cout, _ptr(y), _ptr(ws), ws.numel(), _stream())
    assert rc == 0, L.pdhip_last_error()
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 1e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("N,T,Cc,D", [(1, 64, 128, 64), (2, 256, 128, 32), (1, 1024, 512, 64), (4, 1024, 512, 64), (1, 256, 1024, 64), (2, 64, 1024, 64), (2, 256, 128, 64), (3, 128, 64, 64), (1, 128, 512, 64), (2, 384, 256, 64)])
def test_attention_vs_torch(nn, N, T, Cc, D):
    L = nn['L']
    g = torch.Generator().manual_seed(T + Cc)
    qkv = (torch.randn((N, 3 * Cc, T), generator=g) * 1.5).half().float()
    heads = Cc // D
    q, k, v = qkv.reshape(N * heads, 3 * D, T).split(D, dim=1)
    scale = 1 / math.sqrt(math.sqrt(D))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale), dim=-1)
    ref = torch.einsum("bts,bcs->bct", w, v).reshape(N, Cc, T)
    # a spiked key row forces a large running-max jump in the online softmax (rare-branch test)
    qd = qkv.permute(0, 2, 1).contiguous().half().to(DEV)
    out = torch.empty((N, T, Cc), dtype=torch.float16, device=DEV)
    assert L.pdhip_attention_f16(_ptr(qd), _ptr(out), N, T, Cc, D, None, _stream()) == 0, L.pdhip_last_error()
    out2 = torch.empty_like(out)                          # transposed-V kernel (T % 128 == 0, D == 64), same reference
    vt = torch.empty((N, T, Cc), dtype=torch.float16, device=DEV)
    outs = []
    # K / V chunks requested one or two iterations ahead; V read with the LDS transpose read or from a transposed workspace:
    # the same operands in the same order -> the same bits
    for nbuf, vt_form, qt in ((2, 0, 0), (3, 0, 1), (2, 1, 2), (3, 1, 0), (3, 0, 2), (0, 0, 0)):
        L.pdhip_debug_set_attn(nbuf, vt_form, qt)
        out2.zero_()
        assert L.pdhip_attention_f16(_ptr(qd), _ptr(out2), N, T, Cc, D, _ptr(vt), _stream()) == 0, L.pdhip_last_error()
        outs.append(out2.clone())
    for o_ in outs[1:]:
        assert torch.equal(outs[0], o_)
    o = out.float().cpu().permute(0, 2, 1)
    assert (o - ref).abs().max().item() <= 5e-3 * max(1.0, ref.abs().max().item())
    o2 = out2.float().cpu().permute(0, 2, 1)
    assert (o2 - ref).abs().max().item() <= 5e-3 * max(1.0, ref.abs().max().item())


def test_attention_online_softmax_rescale_branch(nn):
    L = nn['L']
    N, T, Cc, D = 1, 256, 64, 64
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn((N, 3 * Cc, T), generator=g) * 0.5)
    qkv[0, D:2 * D, 200] = qkv[0, 0:D, 17] * 6.0        # key 200 (4th chunk) matches query 17 strongly -> max jumps late
    qkv = qkv.half().float()
    q, k, v = qkv.reshape(1, 3 * D, T).split(D, dim=1)
    w = torch.softmax(torch.einsum("bct,bcs->bts", q, k) / math.sqrt(D), dim=-1)
    ref = torch.einsum("bts,bcs->bct", w, v)
    qd = qkv.permute(0, 2, 1).contiguous().half().to(DEV)
    out = torch.empty((N, T, Cc), dtype=torch.float16, device=DEV)
    assert L.pdhip_attention_f16(_ptr(qd), _ptr(out), N, T, Cc, D, None, _stream()) == 0
    o = out.float().cpu().permute(0, 2, 1)
    assert w[0, 17, 200] > 0.5
    assert (o - ref).abs().max().item() <= 5e-3 * max(1.0, ref.abs().max().item())
    vt = torch.empty((N, T, Cc), dtype=torch.float16, device=DEV)      # the transposed-V kernel takes the same late jump
    assert L.pdhip_attention_f16(_ptr(qd), _ptr(out), N, T, Cc, D, _ptr(vt), _stream()) == 0
    o = out.float().cpu().permute(0, 2, 1)
    assert (o - ref).abs().max().item() <= 5e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("T,peak", [(4096, 0.0), (4096, 5.0)])
def test_attention_long_sequence_denominator(nn, T, peak):
    """The T >= 128 kernel takes the softmax denominator from the matrix pipe over the f16-ROUNDED probabilities (the reference,
    unet.py:336-352 QKVAttentionLegacy, normalises in fp32 and rounds afterwards).  A sequence four times the longest of the network,
    flat logits (4 096 nearly equal weights of 2^-12: every rounding enters the normaliser) and sharply peaked ones (one key per
    query carries most of the mass, the rest underflow towards f16 zero) against the f64 oracle, the short sequences' 5e-3 bound;
    both kernels (LDS-transposed V with a workspace handle, and the generic one)."""
    L = nn['L']
    N, Cc, D = 1, 512, 64
    heads = Cc // D
    g = torch.Generator().manual_seed(T + int(peak * 10))
    qkv = torch.randn((N, 3 * Cc, T), generator=g) * (0.05 if peak == 0.0 else 1.0)
    if peak > 0:                                            # key (7 t + 3) % T is aligned with query t in every head
        idx = (7 * torch.arange(T) + 3) % T
        for h in range(heads):
            qkv[0, h * 3 * D + D:h * 3 * D + 2 * D, idx] = qkv[0, h * 3 * D:h * 3 * D + D, :] * peak
    qkv = qkv.half().float()
    ref = torch.empty((N, Cc, T))
    for h in range(heads):
        q, k, v = qkv[0, h * 3 * D:(h + 1) * 3 * D].double().split(D, dim=0)
        w = torch.softmax((q.t() @ k) / math.sqrt(D), dim=-1)
        if peak >= 5.0:
            assert w.max(-1).values.median() > 0.5
        ref[0, h * D:(h + 1) * D] = (w @ v.t()).t().float()
    qd = qkv.permute(0, 2, 1).contiguous().half().to(DEV)
    vt = torch.empty((N, T, Cc), dtype=torch.float16, device=DEV)
    for ws in (vt, None):
        out = torch.zeros((N, T, Cc), dtype=torch.float16, device=DEV)
        assert L.pdhip_attention_f16(_ptr(qd), _ptr(out), N, T, Cc, D, None if ws is None else _ptr(ws), _stream()) == 0, L.pdhip_last_error()
        o = out.float().cpu().permute(0, 2, 1)
        assert torch.isfinite(o).all()
        assert (o - ref).abs().max().item() <= 5e-3 * max(1.0, ref.abs().max().item())


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()


def test_unet_small_vs_oracle_and_reference_golden(nn):
    g = load_golden('unet_small.npz')
    cfg = ounet.make_config(int(g['cfg_image_size']), int(g['cfg_channels']), 2, "32,16,8", int(g['cfg_head']), True)
    w = ounet.random_weights(cfg, int(g['seed']))
    assert len(w) == int(g['n_tensors'])
    m = nn['di'].UNetModel(image_size=cfg['image_size'], num_channels=cfg['model_channels'], num_head_channels=cfg['num_head_channels'],
                           max_batch=2, device=DEV)
    assert m.load_state_dict(w, strict=True) == len(w)
    x, t = torch.from_numpy(g['x']), torch.from_numpy(g['t'])
    out = m(x.to(DEV), t.to(DEV)).cpu()
    ref = torch.from_numpy(g['ref_out'])                            # the imported reference's fp32 output
    linf, l2 = _rel(out, ref)
    note_measured(test='unet_small_fp32', linf=linf, l2=l2)
    assert linf <= U1_SMALL_FP32_LINF and l2 <= U1_SMALL_FP32_L2, (linf, l2)
    taps = {}
    oo = ounet.forward(cfg, w, x, t, taps=taps)
    assert _rel(out, oo)[0] <= U1_SMALL_FP32_LINF
    # batch independence / ragged batch: one image alone gives the same answer
    out1 = m(x[1:2].to(DEV), t[1:2].to(DEV)).cpu()
    # (same f16 arithmetic, but the split-K factor -- hence the f32 summation order -- depends on the batch: f16-ulp differences)
    assert _rel(out1, out[1:2])[0] <= 1e-2 and _rel(out1, out[1:2])[1] <= 2.5e-3, _rel(out1, out[1:2])
    with pytest.raises(nn['lib'].PdhipError):
        m(torch.zeros((3, 3, 64, 64), device=DEV), torch.zeros((3,), device=DEV))      # > max_batch
    with pytest.raises(nn['lib'].PdhipError):
        nn['di'].UNetModel(image_size=64, num_channels=32, num_head_channels=32, max_batch=1, device=DEV).forward(
            x[:1].to(DEV), t[:1].to(DEV))                           # weights never loaded -> loud failure


def test_unet_full_256_vs_reference_golden(nn):
    g = load_golden('unet_full.npz')
    cfg = ounet.make_config(256, 256, 2, "32,16,8", 64, True)
    w = ounet.random_weights(cfg, int(g['seed']))
    assert len(w) == 566 and int(g['n_params']) == 552814086
    m = nn['di'].UNetModel(max_batch=1, device=DEV, **nn['di'].IMAGENET_256)
    m.load_state_dict(w, strict=True)
    del w
    st = int(g['stride'])
    outs = {}
    for fuse, fold in ((0, 1), (1, 1), (0, 0)):             # GroupNorm stand-alone / inside the halo conv; resample folded / as passes
        old, oldf = nn['L'].pdhip_debug_set_fuse_gn(fuse), nn['L'].pdhip_debug_set_fold_resample(fold)
        try:
            out = m(torch.from_numpy(g['x']).to(DEV), torch.from_numpy(g['t']).to(DEV)).cpu()
        finally:
            nn['L'].pdhip_debug_set_fuse_gn(old); nn['L'].pdhip_debug_set_fold_resample(oldf)
        linf, l2 = _rel(out[:, :, ::st, ::st], torch.from_numpy(g['ref_out']))
        note_measured(test='unet_full_fp32', fuse=fuse, fold=fold, linf=linf, l2=l2)
        assert linf <= U1_FP32_LINF and l2 <= U1_FP32_L2, (fuse, fold, linf, l2)
        outs[(fuse, fold)] = out
    assert torch.equal(outs[(0, 1)], outs[(0, 0)]), "folding the resampled x branch into its consumers is bit-neutral"


def test_ddnm_schedule_and_step_vs_reference_sampler_golden(nn):
    L = nn['L']
    g = load_golden('ddnm_sampler.npz')
    at, an, t, tn, co = nn['di'].ddnm_schedule()
    assert np.array_equal(at, g['at']) and np.array_equal(an, g['at_next'])
    assert np.array_equal(t, g['t']) and np.array_equal(tn, g['t_next'])
    masked, mask, tape = torch.from_numpy(g['masked']).to(DEV), torch.from_numpy(g['mask']).to(DEV), torch.from_numpy(g['tape']).to(DEV)
    wk = torch.from_numpy(g['toy_w']).to(DEV)
    HW = masked.shape[-1] ** 2
    y = torch.empty_like(masked)
    assert L.pdhip_ddnm_prepare(_ptr(masked), _ptr(mask), _ptr(y), 1, HW, _stream()) == 0
    x = tape[0].clone().contiguous()
    for k in range(100):
        tt = torch.ones(1, device=DEV) * float(t[k])
        et = (F.conv2d(x, wk, padding=1) * (0.5 + tt.view(-1, 1, 1, 1) / 1000.0)).contiguous()     # the fixture's toy denoiser
        assert L.pdhip_ddnm_step(_ptr(x), _ptr(et), 6, _ptr(y), _ptr(mask), _ptr(tape[k + 1].contiguous()), 0, k, 1, HW, _stream()) == 0
    out = torch.clamp((x + 1) / 2, 0, 1).cpu()
    assert (out - torch.from_numpy(g['ref_out'])).abs().max().item() <= 1e-5     # vs the reference's own sampler


def test_ddnm_sampler_with_unet_vs_oracle(nn):
    g = load_golden('unet_small.npz')
    cfg = ounet.make_config(64, 32, 2, "32,16,8", 32, True)
    w = ounet.random_weights(cfg, 21)
    inp = nn['di'].Inpainter(DEV, ckpt_path=None, model_kwargs=dict(image_size=64, num_channels=32, num_head_channels=32),
                             max_batch=2, state_dict=w)
    gen = torch.Generator().manual_seed(3)
    V, S, steps = 3, 64, 3                                            # V > max_batch exercises chunking
    imgs = torch.rand((V, 3, S, S), generator=gen)
    masks = (torch.rand((V, S, S), generator=gen) > 0.7).float()
    imgs = imgs * masks[:, None]
    xT = torch.randn((V, 3, S, S), generator=gen)
    tape = torch.randn((steps, V, 3, S, S), generator=gen)
    out = inp.inpaint_views(imgs.to(DEV), masks.to(DEV), x_T=xT.to(DEV), eps_tape=tape.to(DEV), n_steps=steps).cpu()
    ref = oddnm.sample(lambda x, t: ounet.forward(cfg, w, x, t), imgs, masks, xT, list(tape), n_steps=steps)
    assert (out - ref).abs().max().item() <= 3e-2
    # known pixels are reproduced by the masked projection up to the remaining noise level only after many steps;
    # what must hold exactly is determinism and the reference-signature wrapper
    out2 = inp.inpaint_views(imgs.to(DEV), masks.to(DEV), x_T=xT.to(DEV), eps_tape=tape.to(DEV), n_steps=steps).cpu()
    assert torch.equal(out, out2)
    one = inp.inpaint(imgs[:1].permute(0, 2, 3, 1).to(DEV), masks[:1, :, :, None].repeat(1, 1, 1, 3).to(DEV))
    assert one.shape == (1, 3, S, S) and float(one.min()) >= 0 and float(one.max()) <= 1


def test_philox_normal_stream(nn):
    L = nn['L']
    n = 1 << 20
    a = torch.empty((n,), device=DEV)
    b = torch.empty((n,), device=DEV)
    assert L.pdhip_philox_normal(_ptr(a), n, 1234, 1, _stream()) == 0
    assert L.pdhip_philox_normal(_ptr(b), n, 1234, 1, _stream()) == 0
    assert torch.equal(a, b)
    assert L.pdhip_philox_normal(_ptr(b), n, 1234, 2, _stream()) == 0
    assert not torch.equal(a, b)
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1) < 5e-3
    assert abs((a * b).mean().item()) < 5e-3
    assert abs((a ** 4).mean().item() - 3.0) < 0.1


@pytest.mark.parametrize("N,H,W,Cin,Cout,res,splits", [
    (2, 8, 64, 64, 128, False, 0), (1, 16, 64, 96, 192, True, 0), (1, 4, 128, 32, 128, False, 0), (2, 8, 128, 64, 256, True, 0),
    (1, 2, 256, 64, 128, False, 0), (1, 6, 256, 32, 256, True, 0), (1, 8, 256, 160, 128, False, 0),
    (2, 32, 32, 96, 128, True, 0), (1, 16, 32, 64, 256, False, 2), (2, 32, 32, 160, 128, True, 3), (1, 16, 64, 128, 128, True, 4)])
def test_conv3x3_halo_kernel_vs_torch_fp32(nn, N, H, W, Cin, Cout, res, splits):
    """The halo-resident 3x3 kernel (nn_conv_halo.hip; W = 32 / 64 / 128 / 256, 512-pixel tiles): image borders, tile borders
    inside an image, several images, odd / single channel-chunk counts, residual, and the split over channel chunks (f32
    partials + fixed-order reduce) used by small-M layers."""
    L = nn['L']
    old = L.pdhip_debug_set_conv_tile(32)
    ws = torch.empty((max(splits, 1) * N * H * W * Cout,), device=DEV)
    L.pdhip_debug_set_conv_splitk(_ptr(ws), ws.numel(), splits)
    try:
        g = torch.Generator().manual_seed(N * 1000 + H * W + Cin + Cout)
        x = torch.randn((N, Cin, H, W), generator=g).half().float()
        w = (torch.randn((Cout, Cin, 3, 3), generator=g) / math.sqrt(Cin * 9)).half().float()
        b = (torch.randn((Cout,), generator=g) * 0.1).half().float()
        r = torch.randn((N, Cout, H, W), generator=g).half().float() if res else None
        ref = F.conv2d(x, w, b, padding=1).half().float()
        if res:
            ref = (ref + r).half().float()
        out = hip_conv(nn, x, w, b, r)
        assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-3
    finally:
        L.pdhip_debug_set_conv_tile(old)
        L.pdhip_debug_set_conv_splitk(None, 0, 0)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("N,H,Cin,Cout,res", [(2, 16, 64, 128, True), (1, 24, 96, 256, False)])
def test_conv3x3_halo_tile_geometries_on_256_wide_images(nn, mode, N, H, Cin, Cout, res):
    """256-wide images: tiles of 4 rows x 128 columns (automatic) or full rows (1) -- strip borders inside
    the image must read their neighbours' columns, the image border the zero padding; the fused GroupNorm partials (one chunk per
    tile) are covered by the UNet tests, which run the automatic geometry."""
    L = nn['L']
    old = (L.pdhip_debug_set_conv_tile(32), L.pdhip_debug_set_conv_halo_strips(mode))
    try:
        g = torch.Generator().manual_seed(7 * H + Cin + mode)
        x = torch.randn((N, Cin, H, 256), generator=g).half().float()
        w = (torch.randn((Cout, Cin, 3, 3), generator=g) / math.sqrt(Cin * 9)).half().float()
        b = (torch.randn((Cout,), generator=g) * 0.1).half().float()
        r = torch.randn((N, Cout, H, 256), generator=g).half().float() if res else None
        ref = F.conv2d(x, w, b, padding=1).half().float()
        if res:
            ref = (ref + r).half().float()
        out = hip_conv(nn, x, w, b, r)
        assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-3
    finally:
        L.pdhip_debug_set_conv_tile(old[0]); L.pdhip_debug_set_conv_halo_strips(old[1])


@pytest.mark.parametrize("bk,stages,wmw", [(32, 2, 2), (32, 3, 2), (32, 4, 4), (64, 2, 2), (64, 3, 4), (64, 2, 4), (32, 3, 4),
                                           (64, 2, 8), (64, 12, 2), (64, 12, 8), (64, 12, 16)])
def test_conv_igemm_all_kernel_variants(nn, bk, stages, wmw):
    """Every (K-step, pipeline depth, tile height) instantiation of the conv kernel computes the same convolution
    (stages 12 = two LDS stages with the hand-scheduled register-pipelined fragment loop)."""
    L = nn['L']
    old = (L.pdhip_debug_set_conv_bk(bk), L.pdhip_debug_set_conv_stages(stages), L.pdhip_debug_set_conv_tile(wmw))
    try:
        for (N, H, W, Cin, Cout, k, res) in [(2, 16, 16, 128, 192, 3, True), (1, 24, 24, 64, 128, 1, False), (3, 8, 8, 256, 64, 3, False),
                                             (2, 16, 16, 128, 256, 3, True), (1, 20, 20, 64, 512, 1, False), (1, 32, 32, 64, 256, 3, False)]:
            g = torch.Generator().manual_seed(bk + stages + wmw + Cin)
            x = torch.randn((N, Cin, H, W), generator=g).half().float()
            w = (torch.randn((Cout, Cin, k, k), generator=g) / math.sqrt(Cin * k * k)).half().float()
            b = (torch.randn((Cout,), generator=g) * 0.1).half().float()
            r = torch.randn((N, Cout, H, W), generator=g).half().float() if res else None
            ref = F.conv2d(x, w, b, padding=k // 2).half().float()
            if res:
                ref = (ref + r).half().float()
            out = hip_conv(nn, x, w, b, r)
            assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-3
    finally:
        L.pdhip_debug_set_conv_bk(old[0]); L.pdhip_debug_set_conv_stages(old[1]); L.pdhip_debug_set_conv_tile(old[2])


@pytest.mark.parametrize("N,H,W,Cin,Cout,film,res,splits", [
    (2, 16, 32, 64, 128, False, False, 0), (1, 32, 32, 96, 256, True, True, 0), (2, 8, 64, 64, 128, True, False, 0),
    (1, 16, 64, 160, 128, False, True, 0), (2, 8, 128, 64, 256, True, True, 0), (1, 4, 128, 32, 128, False, False, 0),
    (1, 4, 256, 64, 128, True, False, 0), (2, 8, 256, 96, 256, False, True, 0), (1, 16, 64, 128, 128, True, True, 2),
    (3, 16, 32, 64, 128, True, False, 3)])
def test_gn_silu_fused_into_halo_conv_vs_torch_fp32(nn, N, H, W, Cin, Cout, film, res, splits):
    """The APPLY variant of the halo-resident 3x3 kernel: y = conv3x3(silu(GroupNorm32(x) [* (1 + scale) + shift])) (+ residual)
    with the normalisation applied in LDS while the input tile is staged (in_layers / out_layers of a ResBlock, unet.py:183-252)
    against plain torch fp32 on the same f16-rounded inputs: image borders (zero padding of the TRANSFORMED image), tile borders,
    several images with different statistics and FiLM rows, odd chunk counts, residual, the split over channel chunks."""
    L = nn['L']
    g = torch.Generator().manual_seed(N * 1000 + H + W + Cin)
    x = (torch.randn((N, Cin, H, W), generator=g) * (1.0 + torch.rand((N, Cin, 1, 1), generator=g)) + torch.randn((N, Cin, 1, 1), generator=g)).half().float()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / math.sqrt(9 * Cin)).half().float()
    b = (torch.randn((Cout,), generator=g) * 0.1).half().float()
    gamma = 1.0 + 0.2 * torch.randn((Cin,), generator=g)
    beta = 0.2 * torch.randn((Cin,), generator=g)
    fl = (0.3 * torch.randn((N, 2 * Cin), generator=g)) if film else None
    r = torch.randn((N, Cout, H, W), generator=g).half().float() if res else None
    # fp32 reference
    h = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    if film:
        h = h * (1 + fl[:, :Cin, None, None].half().float()) + fl[:, Cin:, None, None].half().float()
    ref = F.conv2d(F.silu(h), w, b, padding=1)
    if res:
        ref = ref + r
    pad = ((Cout + 127) // 128) * 128
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    wp = torch.zeros((pad, 9 * Cin), dtype=torch.float16, device=DEV)
    assert L.pdhip_pack_conv_weight_f16(_ptr(w.to(DEV)), Cout, Cin, 9, _ptr(wp), _stream()) == 0
    bd, gd, be = b.to(DEV), gamma.to(DEV), beta.to(DEV)
    fd = fl.to(DEV).contiguous() if film else None
    rd = r.permute(0, 2, 3, 1).contiguous().half().to(DEV) if res else None
    y = torch.empty((N, H, W, Cout), dtype=torch.float16, device=DEV)
    zp = torch.zeros((128,), dtype=torch.float16, device=DEV)
    nws = N * 64 + N * 64 * ((H * W + 255) // 256) + N * Cin * 2
    ws = torch.empty((nws,), device=DEV)
    skws = torch.empty((max(splits, 1) * N * H * W * Cout,), device=DEV)
    old = L.pdhip_debug_set_conv_splitk(_ptr(skws), skws.numel(), splits)
    try:
        rc = L.pdhip_gn_silu_conv3x3_nhwc_f16(_ptr(xd), _ptr(gd), _ptr(be), _ptr(fd) if film else None, 2 * Cin, _ptr(wp), _ptr(bd),
                                              _ptr(rd) if res else None, _ptr(y), N, H, W, Cin, Cout, pad, _ptr(zp), _ptr(ws), nws, _stream())
    finally:
        L.pdhip_debug_set_conv_splitk(None, 0, 0)
    assert rc == 0, L.pdhip_last_error()
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 3e-3, err
    # and it agrees with the two-pass form (stand-alone GroupNorm kernel, then the plain conv) to f16 rounding
    hn = torch.empty((N, H, W, Cin), dtype=torch.float16, device=DEV)
    st = torch.empty((N * 64,), device=DEV)
    gws = torch.empty((N * 64 * ((H * W + 255) // 256),), device=DEV)
    assert L.pdhip_groupnorm_nhwc_f16(_ptr(xd), _ptr(gd), _ptr(be), _ptr(fd) if film else None, N, H, W, Cin, 1, 0, _ptr(hn), _ptr(st),
                                      _ptr(gws), gws.numel(), _stream()) == 0
    y2 = torch.empty_like(y)
    assert L.pdhip_conv2d_nhwc_f16(_ptr(hn), _ptr(wp), _ptr(bd), _ptr(rd) if res else None, _ptr(y2), N, H, W, Cin, Cout, pad, 9, _ptr(zp),
                                   _stream()) == 0
    torch.cuda.synchronize()
    assert (y.float() - y2.float()).abs().max().item() / ref.abs().max().item() <= 3e-3
