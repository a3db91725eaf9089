"""Round-5 GPU parity tests: U1 pinned to the reference's OWN fp16 forward (configs/imagenet_256.yml:26 use_fp16: true,
diffusion.py:438-439 convert_to_fp16(), unet.py:619-625 / 655-663, fp16_util.py:15-22, nn.py:17-19) -- fixtures
tests/golden/unet_small_fp16.npz / unet_full_fp16.npz written by tools/gen_golden_nn.py small16 / full16 from the imported reference
run on the CPU -- including the f16 RANGE behaviour: a large-activation weight set (residual stream at 3-5e4, f16's top binade) and one
that overflows one image of a batch (inf -> NaN in the reference; the engine must lose exactly that image and keep the other)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import unet as ounet

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
MEASURED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r05_u1_measured.jsonl')

# Bounds = 2x the largest value measured on the GPU boxes of the round (profiles/r05_u1_bounds.txt); relative L-inf / relative L2.
FP16_SMALL = (4.2e-3, 3.4e-3)          # engine vs the reference's fp16 forward, 64^2 / 32-channel config; measured 2.06e-3 / 1.68e-3
FP16_FULL = (2.3e-3, 2.4e-3)           # ... full 256^2 / 552.8 M-parameter config, batch 1 and 8; measured 1.15e-3 / 1.20e-3
FP16_LARGE_SMALL = (5.5e-3, 4.7e-3)    # ... large-activation weight set (x 4096: peak |activation| 3.5e4), small config; measured 2.74e-3 / 2.32e-3
FP16_LARGE_FULL = (3.8e-3, 3.3e-3)     # ... (x 2048 / x 4096: peaks 2.1e4 / 4.3e4), full config; measured 1.86e-3 / 1.63e-3


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()


def _note(**kw):
    os.makedirs(os.path.dirname(MEASURED), exist_ok=True)
    with open(MEASURED, 'a') as f:
        f.write(json.dumps(kw) + '\n')


@pytest.fixture(scope="module")
def nn():
    assert torch.cuda.is_available()
    from pointdreamer_amd import _lib
    import pointdreamer_amd.ddnm_inpainting as di
    return dict(L=_lib.lib(), lib=_lib, di=di)


def test_unet_small_vs_reference_fp16_forward_and_f16_range(nn):
    g16, g = load_golden('unet_small_fp16.npz'), load_golden('unet_small.npz')
    cfg = ounet.make_config(int(g['cfg_image_size']), int(g['cfg_channels']), 2, "32,16,8", int(g['cfg_head']), True)
    w = ounet.random_weights(cfg, int(g16['seed']))
    x, t = torch.from_numpy(g['x']).to(DEV), torch.from_numpy(g['t']).to(DEV)
    kw = dict(image_size=cfg['image_size'], num_channels=cfg['model_channels'], num_head_channels=cfg['num_head_channels'], max_batch=2, device=DEV)
    # (the reference's fp16 forward sits this far from its own fp32 forward: the engine must be at most as far from the fp16 one)
    assert float(g16['fp16_vs_fp32_linf']) < 2.5e-3 and float(g16['fp16_vs_fp32_l2']) < 2.5e-3
    for factor in [1.0] + [float(b) for b in g16['boosts']]:
        tag = '' if factor == 1.0 else '_x%d' % int(factor)
        m = nn['di'].UNetModel(**kw)
        m.load_state_dict(w if factor == 1.0 else ounet.boost_out_layers(w, factor), strict=True)
        out = m(x, t).cpu()
        ref, fin = torch.from_numpy(g16['ref16_out' + tag]), g16['finite' + tag]
        mine_fin = torch.isfinite(out).flatten(1).all(1).numpy()
        assert np.array_equal(mine_fin, fin), (factor, mine_fin, fin)     # the SAME images survive / overflow as in the reference
        if not fin.all():
            assert not torch.isfinite(out[~torch.from_numpy(fin)]).any(), "an overflowed image is lost whole (inf -> GroupNorm -> NaN), as in the reference"
            assert float(g16['peak' + tag]) > 4.0e4
        for b in np.nonzero(fin)[0]:
            linf, l2 = _rel(out[b:b + 1], ref[b:b + 1])
            _note(test='small_fp16', factor=factor, image=int(b), linf=linf, l2=l2, peak=float(g16['peak' + tag]))
            bound = FP16_SMALL if factor == 1.0 else FP16_LARGE_SMALL
            assert linf <= bound[0] and l2 <= bound[1], (factor, b, linf, l2)


@pytest.mark.parametrize("N", [1, 8])
def test_unet_full_vs_reference_fp16_forward(nn, N):
    g16, g = load_golden('unet_full_fp16.npz'), load_golden('unet_full.npz')
    cfg = ounet.make_config(256, 256, 2, "32,16,8", 64, True)
    w = ounet.random_weights(cfg, int(g16['seed']))
    st = int(g16['stride'])
    gen = torch.Generator().manual_seed(500 + N)
    x = torch.randn((N, 3, 256, 256), generator=gen)
    t = torch.randint(0, 1000, (N,), generator=gen).float()
    x[0], t[0] = torch.from_numpy(g['x'])[0], torch.from_numpy(g['t'])[0]          # image 0 = the fixture's input, the rest fill the batch
    x, t = x.to(DEV), t.to(DEV)
    for factor in [1.0] + [float(b) for b in g16['boosts']]:
        tag = '' if factor == 1.0 else '_x%d' % int(factor)
        m = nn['di'].UNetModel(max_batch=N, device=DEV, **nn['di'].IMAGENET_256)
        m.load_state_dict(w if factor == 1.0 else ounet.boost_out_layers(w, factor), strict=True)
        out = m(x, t)[:1, :, ::st, ::st].cpu()
        del m
        assert bool(g16['finite' + tag][0]) and torch.isfinite(out).all()
        linf, l2 = _rel(out, torch.from_numpy(g16['ref16_out' + tag]))
        _note(test='full_fp16', batch=N, factor=factor, linf=linf, l2=l2, peak=float(g16['peak' + tag]))
        bound = FP16_FULL if factor == 1.0 else FP16_LARGE_FULL
        assert linf <= bound[0] and l2 <= bound[1], (N, factor, linf, l2)
        if factor == 1.0:
            l32 = _rel(out, torch.from_numpy(g['ref_out']))
            _note(test='full_fp32', batch=N, linf=l32[0], l2=l32[1])


# D1 o U1 against the reference's sampler driving its OWN fp16 UNet (tools/gen_golden_nn.py ddnm_full16): bounds = 2x measured
DRIFT16_L2, DRIFT16_LINF = 6.5e-4, 7.0e-4        # x sqrt(k + 1); measured 3.1e-4 / 3.4e-4 at batch 1 and 8 (profiles/r05_u1_bounds.txt)


@pytest.mark.parametrize("batch", [1, 8])
def test_ddnm_unet_full_multistep_vs_reference_sampler_with_its_fp16_unet(nn, batch):
    """The reference's simplified_ddnm_inpainting (diffusion.py:459-570) driving the reference's fp16 UNetModel (use_fp16: true +
    convert_to_fp16(), the configuration it ships: configs/imagenet_256.yml:26) for the first 10 steps of the schedule on two images;
    the engine runs the same steps with the fixture's noise tape at UNet batch 1 and 8."""
    import ctypes as C
    from tools.gen_golden_nn import ddnm_full_inputs
    L = nn['L']
    P = lambda t_: C.c_void_p(t_.data_ptr())
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = load_golden('ddnm_unet_full_fp16.npz')
    steps, n_img, st = int(g['steps']), int(g['n_img']), int(g['stride'])
    cfg = ounet.make_config(256, 256, 2, "32,16,8", 64, True)
    w = ounet.random_weights(cfg, int(g['weight_seed']))
    m = nn['di'].UNetModel(max_batch=batch, device=DEV, **nn['di'].IMAGENET_256)
    m.load_state_dict(w, strict=True)
    del w
    masked, masks, tape = ddnm_full_inputs(int(g['seed']), n_img, steps)
    sel = [i % n_img for i in range(batch)]
    mk = torch.from_numpy(masked[sel]).to(DEV).contiguous()
    ms = torch.from_numpy(masks[sel]).to(DEV).contiguous()
    tp = torch.from_numpy(tape[sel]).to(DEV)
    HW = 256 * 256
    y = torch.empty_like(mk)
    assert L.pdhip_ddnm_prepare(P(mk), P(ms), P(y), batch, HW, S()) == 0
    x = tp[:, 0].clone().contiguous()
    _, _, t_sched, _, _ = nn['di'].ddnm_schedule()
    worst = [0.0, 0.0]
    for k in range(steps):
        tt = torch.full((batch,), float(t_sched[k]), device=DEV)
        et = m(x, tt)
        eps = tp[:, k + 1].contiguous()
        assert L.pdhip_ddnm_step(P(x), P(et), 6, P(y), P(ms), P(eps), 0, k, batch, HW, S()) == 0, L.pdhip_last_error()
        xs = x[:, :, ::st, ::st].cpu()
        for b in range(batch):
            linf, l2 = _rel(xs[b], torch.from_numpy(g['xs'][sel[b], k]))
            worst = [max(worst[0], linf / np.sqrt(k + 1)), max(worst[1], l2 / np.sqrt(k + 1))]
            assert l2 <= DRIFT16_L2 * np.sqrt(k + 1) and linf <= DRIFT16_LINF * np.sqrt(k + 1), (batch, b, k, linf, l2)
    _note(test='ddnm_full_fp16_drift_per_sqrt_k', batch=batch, linf=worst[0], l2=worst[1])


# ---- Uq1-Uq4 with the view loop unrolled at V = 8 and the exponential-free view selection (csrc/unproject.hip, round 5)
@pytest.mark.parametrize("complete", [True, False])
def test_unproject_unrolled_views_equal_the_generic_kernels_incl_near_ties(complete):
    """The V = 8 kernels against the run-time-V kernels (pdhip_debug_set_unproject_generic) and against the oracle at BASELINE sizes, on
    face normals built to sit ON the decision boundary of the view selection: normal = bisector of two view directions (+ a push of
    0, 1e-7 .. 1e-4 towards one of them), zero and NaN normals -- the cases where 'first maximum of the similarity' and 'first maximum of
    the rounded softmax weight' could differ and the kernel must fall back to the generic expressions."""
    import pointdreamer_amd.ours_utils as ou
    import pointdreamer_amd.unproject as up
    import pointdreamer_amd.camera_utils as cu
    from pointdreamer_amd import synthetic, _lib
    from oracle import camera as ocam, unproject as ounp
    L = _lib.lib()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    N_ = lambda t: t.detach().cpu().numpy()
    sh = synthetic.make_shape(30000, 1024)
    cams, base_dirs, eyes, ups = cu.create_cameras(8, 1.6, 512, device=DEV)
    hard, fidx, depth, vuv, uvc, uvs, pad, puv, pdep = ou.get_rendered_hard_mask_and_face_idx_batch(
        cams, T(sh['vertices']), T(sh['faces']), T(sh['points']), None, True, 0.05)
    rng = np.random.default_rng(77)
    inp = torch.from_numpy(rng.random((8, 3, 256, 256), dtype=np.float32)).to(DEV)
    sf = torch.ones(8, device=DEV)
    bd = N_(base_dirs).astype(np.float64)
    F = sh['f_normals'].shape[0]
    fn = sh['f_normals'].copy()
    a, b = rng.integers(0, 8, F), rng.integers(0, 8, F)
    push = rng.choice([0.0, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4], F)
    adv = bd[a] + bd[b] + push[:, None] * bd[b]
    nrm = np.linalg.norm(adv, axis=1, keepdims=True)
    adv = np.where(nrm > 0, adv / np.maximum(nrm, 1e-30), adv)
    pick = rng.random(F) < 0.7
    fn[pick] = adv[pick].astype(np.float32)
    fn[rng.random(F) < 0.01] = 0.0
    fn[rng.random(F) < 0.01] = np.nan
    #  unnormalised normals: |n| = 30 ... 3000 puts sim - max below the f32 exp underflow for the far views -- the reference's
    # softmax weights are then denormal / 0 and its argmax takes the FIRST such candidate, which the similarity shortcut must not override
    big = rng.random(F) < 0.15
    fn[big] = (fn[big] * rng.choice([30.0, 60.0, 90.0, 110.0, 300.0, 3000.0], (int(big.sum()), 1))).astype(np.float32)
    args = (inp, T(fn), 256, cams, 512, base_dirs, T(sh['gb_pos']), T(sh['mask']), T(sh['per_atlas_pixel_face_id']), uvc, uvs, pad, sf, depth,
            [21, 11], complete)
    fast = [N_(t) for t in up.unproject_dense(*args)]
    old = L.pdhip_debug_set_unproject_generic(1)
    try:
        gen = [N_(t) for t in up.unproject_dense(*args)]
    finally:
        L.pdhip_debug_set_unproject_generic(old)
    for x, y, name in zip(fast, gen, ('atlas', 'shrinked', 'view_ids', 'painted', 'visibility')):
        assert np.array_equal(x, y, equal_nan=True), name
    m = sh['mask'][0, :, :, 0]
    assert len(np.unique(fast[2][m])) >= 8
    ocams = [ocam.Camera(N_(c.params), 512) for c in cams]
    o = ounp.unproject(N_(inp), fn, 256, ocams, 512, N_(base_dirs), sh['gb_pos'], sh['mask'], sh['per_atlas_pixel_face_id'], N_(uvc), N_(uvs), pad,
                       N_(sf), N_(depth), [21, 11], complete)
    assert np.array_equal(fast[1], o['shrinked'])
    bad = fast[2][m] != o['point_view_ids']
    if bad.any():                                                    # which kind of face disagrees
        fb = sh['per_atlas_pixel_face_id'].reshape(-1)[m.reshape(-1)][bad]
        kinds = dict(nan=int(np.isnan(fn[fb]).any(1).sum()), zero=int((fn[fb] == 0).all(1).sum()), adversarial=int(pick[fb].sum()), n=int(bad.sum()))
        raise AssertionError(f"view ids differ from the oracle: {kinds}, pushes {np.unique(push[fb], return_counts=True)}")
    assert np.array_equal(fast[0], o['atlas_img'])
