"""Golden vectors for rows U1 and D1, produced by RUNNING THE REFERENCE's own modules on the CPU.

  python -m tools.gen_golden_nn            (build container only)

U1: the reference UNetModel (models/DDNM/guided_diffusion/unet.py, built by script_util.create_model) is
    imported unmodified, loaded (strict) with oracle.unet.random_weights(cfg, seed) and run in fp32:
      unet_small.npz   image 64, 32 channels (channel_mult (1,2,3,4), attention at ds 2,4,8, 32-ch heads)
      unet_full.npz    the real 256x256 configuration of configs/imagenet_256.yml (552.8 M params); only the
                       input, t and a strided subsample of the output are stored.
D1: the reference Diffusion.simplified_ddnm_inpainting (diffusion.py:459-570) itself, with `.to('cuda')`
    redirected to the CPU and torch.randn / randn_like replaced by a recorded noise tape (the tape is the
    injected noise of the fixture), driving a tiny deterministic stand-in denoiser.
"""
import os
import sys
import types
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import ref_harness as rh            # noqa: E402
from oracle import unet as ounet, ddnm as oddnm  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def ref_unet(cfg_kwargs, weights):
    unet_mod, script_util, gnn = rh.import_reference_unet()
    model = script_util.create_model(**cfg_kwargs)
    missing = model.load_state_dict(weights, strict=True)
    model.eval()
    return model


SMALL = dict(image_size=64, num_channels=32, num_res_blocks=2, attention_resolutions="32,16,8", num_head_channels=32,
             learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True, use_fp16=False, num_heads=4,
             num_heads_upsample=-1, class_cond=False, use_checkpoint=False, dropout=0.0,
             use_new_attention_order=False, channel_mult="")
FULL = dict(SMALL, image_size=256, num_channels=256, num_head_channels=64)


def gen_unet(name, kw, seed, batch, stride):
    cfg = ounet.make_config(kw['image_size'], kw['num_channels'], kw['num_res_blocks'], kw['attention_resolutions'],
                            kw['num_head_channels'], kw['learn_sigma'])
    w = ounet.random_weights(cfg, seed)
    model = ref_unet(kw, w)
    assert len(w) == len(model.state_dict())
    g = torch.Generator().manual_seed(seed + 1)
    S = kw['image_size']
    x = torch.randn((batch, 3, S, S), generator=g)
    t = torch.tensor([990.0, 350.0, 0.0][:batch])
    with torch.no_grad():
        y = model(x, t)
        yo = ounet.forward(cfg, w, x, t)
    err = (y - yo).abs().max().item()
    print(name, 'reference vs oracle max abs diff', err, 'out std', y.std().item())
    assert err < 2e-4 * max(1.0, y.abs().max().item())
    np.savez_compressed(os.path.join(OUT, name), seed=seed, x=x.numpy(), t=t.numpy(), stride=stride,
                        ref_out=y.numpy()[:, :, ::stride, ::stride].copy(), n_tensors=len(w),
                        n_params=sum(v.numel() for v in w.values()), cfg_image_size=S, cfg_channels=kw['num_channels'],
                        cfg_head=kw['num_head_channels'])


def ref_unet_fp16(cfg_kwargs, weights):
    """The reference's OWN fp16 path (configs/imagenet_256.yml:26 use_fp16: true; diffusion.py:438-439): create_model(use_fp16=True),
    strict load of the f32 weights, convert_to_fp16() (unet.py:619-625 -> fp16_util.py:15-22: conv weights and biases of the torso to
    half), forward on the CPU (h = x.type(self.dtype) ... h.type(x.dtype), unet.py:655-663; GroupNorm32 and softmax in f32, nn.py:17-19,
    unet.py:352)."""
    unet_mod, script_util, gnn = rh.import_reference_unet()
    model = script_util.create_model(**dict(cfg_kwargs, use_fp16=True))
    model.load_state_dict(weights, strict=True)
    model.convert_to_fp16()
    model.eval()
    assert model.dtype == torch.float16 and model.input_blocks[0][0].weight.dtype == torch.float16 and model.out[2].weight.dtype == torch.float32
    return model


def gen_unet_fp16(name, kw, seed, batch, stride, src, boosts=()):
    """U1 pinned to the reference's fp16 forward: same seeded weights and the same x, t as `src` (the fp32 fixture),
    outputs of the fp16 model; `boosts`: large-activation variants (oracle.unet.boost_out_layers) -- per factor the output, the
    per-image finiteness and the largest finite |activation| any conv produced."""
    import time
    cfg = ounet.make_config(kw['image_size'], kw['num_channels'], kw['num_res_blocks'], kw['attention_resolutions'],
                            kw['num_head_channels'], kw['learn_sigma'])
    w = ounet.random_weights(cfg, seed)
    g = np.load(os.path.join(OUT, src))
    x, t = torch.from_numpy(g['x']), torch.from_numpy(g['t'])
    assert int(g['seed']) == seed and x.shape[0] == batch
    out = dict(seed=seed, stride=stride, src=src)
    for factor in (1.0,) + tuple(boosts):
        wb = w if factor == 1.0 else ounet.boost_out_layers(w, factor)
        model = ref_unet_fp16(kw, wb)
        peak = {'v': 0.0}

        def hook(mod, inp, o):
            if torch.is_tensor(o):
                f = o.float().abs()
                f = f[torch.isfinite(f)]
                if f.numel():
                    peak['v'] = max(peak['v'], float(f.max()))
        for mod in model.modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.Conv1d)):
                mod.register_forward_hook(hook)
        t0 = time.time()
        with torch.no_grad():
            y = model(x, t)
        fin = torch.isfinite(y).flatten(1).all(1)
        tag = '' if factor == 1.0 else '_x%d' % int(factor)
        print(name, 'factor', factor, '%.0f s' % (time.time() - t0), 'finite per image', fin.tolist(), 'peak finite |conv out| %.0f' % peak['v'],
              'out std', float(y[fin].std()) if fin.any() else None, flush=True)
        if factor == 1.0:
            ref32 = torch.from_numpy(g['ref_out'])
            d = y[:, :, ::stride, ::stride] - ref32
            out['fp16_vs_fp32_linf'] = float(d.abs().max() / ref32.abs().max())
            out['fp16_vs_fp32_l2'] = float(d.norm() / ref32.norm())
            print(name, 'reference fp16 vs reference fp32: rel Linf %.3e rel L2 %.3e' % (out['fp16_vs_fp32_linf'], out['fp16_vs_fp32_l2']))
        out['ref16_out' + tag] = np.nan_to_num(y.numpy()[:, :, ::stride, ::stride], nan=0.0, posinf=0.0, neginf=0.0).copy()
        out['finite' + tag] = fin.numpy()
        out['peak' + tag] = peak['v']
    out['boosts'] = np.array(boosts, np.float64)
    np.savez_compressed(os.path.join(OUT, name), **out)


def gen_ddnm(name):
    rh.install()
    # the reference's diffusion.py pulls torchvision.utils and its datasets package; both are stubbed/importable
    import importlib
    ds = types.ModuleType('datasets_stub')
    cwd = os.getcwd()
    os.chdir(rh.REF)
    try:
        sys.path.insert(0, os.path.join(rh.REF, 'models', 'DDNM'))
        diffusion = importlib.import_module('models.DDNM.guided_diffusion.diffusion')
    finally:
        os.chdir(cwd)
    import yaml

    class NS(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)
        __setattr__ = dict.__setitem__

    def to_ns(d):
        return NS({k: to_ns(v) if isinstance(v, dict) else v for k, v in d.items()})
    config = to_ns(yaml.safe_load(open(os.path.join(rh.REF, 'models/DDNM/configs/imagenet_256.yml'))))
    args = NS(sigma_y=0, eta=0.85, seed=1234)
    runner = diffusion.Diffusion(args, config, device=torch.device('cpu'))
    H = 32
    config.data.image_size = H
    g = torch.Generator().manual_seed(77)
    masked = torch.rand((1, 3, H, H), generator=g)
    mask = (torch.rand((1, H, H), generator=g) > 0.6).float()
    masked = masked * mask[:, None]
    tape = [torch.randn((1, 3, H, H), generator=g) for _ in range(101)]
    pos = {'i': 0}

    def next_noise(*a, **k):
        n = tape[pos['i']]
        pos['i'] += 1
        return n.clone()
    wk = torch.randn((6, 3, 3, 3), generator=g) * 0.2

    def toy_model(x, t):
        return torch.nn.functional.conv2d(x, wk, padding=1) * (0.5 + t.view(-1, 1, 1, 1) / 1000.0)
    orig_to, orig_randn, orig_randn_like = torch.Tensor.to, torch.randn, torch.randn_like

    def to(self, *a, **k):
        if a and a[0] == 'cuda':
            return self
        return orig_to(self, *a, **k)
    torch.Tensor.to, torch.randn, torch.randn_like = to, next_noise, next_noise
    try:
        out = runner.simplified_ddnm_inpainting(toy_model, masked.unsqueeze(0), mask)
    finally:
        torch.Tensor.to, torch.randn, torch.randn_like = orig_to, orig_randn, orig_randn_like
    assert pos['i'] == 101
    out = out[0]
    mine = oddnm.sample(toy_model, masked, mask, tape[0], tape[1:])
    err = (mine - out).abs().max().item()
    print(name, 'reference sampler vs oracle max abs diff', err)
    assert err < 1e-5
    cos = oddnm.step_coefficients()
    np.savez_compressed(os.path.join(OUT, name), masked=masked.numpy(), mask=mask.numpy(), toy_w=wk.numpy(),
                        tape=torch.stack(tape).numpy(), ref_out=out.numpy(),
                        ref_betas=runner.betas.numpy(),
                        at=np.array([c['at'].item() for c in cos], np.float32),
                        at_next=np.array([c['at_next'].item() for c in cos], np.float32),
                        t=np.array([c['t'] for c in cos]), t_next=np.array([c['t_next'] for c in cos]))


def _import_diffusion():
    rh.install()
    import importlib
    cwd = os.getcwd()
    os.chdir(rh.REF)
    try:
        sys.path.insert(0, os.path.join(rh.REF, 'models', 'DDNM'))
        return importlib.import_module('models.DDNM.guided_diffusion.diffusion')
    finally:
        os.chdir(cwd)


def ddnm_full_inputs(seed, n_img, steps, S=256):
    """Inputs of the full-size D1 o U1 fixture, regenerated from the seed by the test (numpy PCG64 streams are stable across
    numpy versions): masked images, masks, x_T and the per-step noise tape, image-major."""
    rng = np.random.default_rng(seed)
    imgs = rng.random((n_img, 3, S, S), dtype=np.float32)
    masks = (rng.random((n_img, S, S), dtype=np.float32) > 0.6).astype(np.float32)
    tape = rng.standard_normal((n_img, steps + 1, 3, S, S), dtype=np.float32)      # [:, 0] = x_T, [:, k + 1] = noise of step k
    return imgs * masks[:, None], masks, tape


def gen_ddnm_full(name, steps=10, n_img=2, seed=2024, stride=4, fp16=False):
    """D1 o U1 at full size: the reference's own simplified_ddnm_inpainting (diffusion.py:459-570) driving the reference's own
    fp32 UNetModel (552.8 M parameters, seeded weights) for the first `steps` of the 100-step schedule; the loop is cut after
    `steps` updates by the model wrapper.  Stored: a strided sample of x_k after every update and the full x after the last."""
    diffusion = _import_diffusion()
    import yaml

    class NS(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)
        __setattr__ = dict.__setitem__

    def to_ns(d):
        return NS({k: to_ns(v) if isinstance(v, dict) else v for k, v in d.items()})
    config = to_ns(yaml.safe_load(open(os.path.join(rh.REF, 'models/DDNM/configs/imagenet_256.yml'))))
    cfg = ounet.make_config(256, 256, 2, "32,16,8", 64, True)
    w = ounet.random_weights(cfg, 12)
    model = ref_unet_fp16(FULL, w) if fp16 else ref_unet(FULL, w)      # fp16: the reference's own fp16 torso, same sampler
    masked, masks, tape = ddnm_full_inputs(seed, n_img, steps)
    xs_all, final_all = [], []

    class Cut(Exception):
        pass
    orig_to, orig_randn, orig_randn_like = torch.Tensor.to, torch.randn, torch.randn_like
    for im in range(n_img):
        args = NS(sigma_y=0, eta=0.85, seed=1234)
        runner = diffusion.Diffusion(args, config, device=torch.device('cpu'))
        pos = {'i': 0}
        seen = []

        def next_noise(*a, **k):
            n = torch.from_numpy(tape[im, pos['i']][None].copy())
            pos['i'] += 1
            return n

        def wrapped(x, t):
            seen.append(x.detach().clone())               # the sampler's state entering this step = x after the previous update
            if len(seen) == steps + 1:
                raise Cut()
            assert float(t[0]) == 990.0 - 10.0 * (len(seen) - 1)
            with torch.no_grad():
                return model(x, t)

        def to(self, *a, **k):
            if a and a[0] == 'cuda':
                return self
            return orig_to(self, *a, **k)
        torch.Tensor.to, torch.randn, torch.randn_like = to, next_noise, next_noise
        try:
            runner.simplified_ddnm_inpainting(wrapped, torch.from_numpy(masked[im:im + 1]).unsqueeze(0), torch.from_numpy(masks[im:im + 1]))
            raise AssertionError('the sampler was expected to be cut')
        except Cut:
            pass
        finally:
            torch.Tensor.to, torch.randn, torch.randn_like = orig_to, orig_randn, orig_randn_like
        assert pos['i'] == steps + 1 and len(seen) == steps + 1
        assert torch.equal(seen[0], torch.from_numpy(tape[im, 0][None]))
        xs = torch.cat(seen[1:], 0).numpy()               # [steps, 3, S, S]: x after update k = 0 .. steps-1
        print(name, 'image', im, 'x_k std', [round(float(v.std()), 4) for v in xs])
        xs_all.append(xs[:, :, ::stride, ::stride].copy())
        final_all.append(xs[-1].copy())
    np.savez_compressed(os.path.join(OUT, name), seed=seed, steps=steps, n_img=n_img, stride=stride, weight_seed=12,
                        xs=np.stack(xs_all), x_last=np.stack(final_all))


def gen_ddnm_full100(name, n_img=2, seed=4100, stride=4, every=10, threads=6):
    """D1 o U1 over the WHOLE 100-step schedule at full size (VERDICT r3 missing 1): the reference's simplified_ddnm_inpainting
    (diffusion.py:459-570) driving the reference's fp32 UNetModel to completion, nothing cut.  Stored: a strided sample of the
    sampler state after updates every-1, 2*every-1, ... (k = 9, 19, .., 99), and the sampler's full return value (the
    inverse_data_transform'ed, clamped image, diffusion.py:563-566)."""
    diffusion = _import_diffusion()
    import yaml
    import time

    class NS(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)
        __setattr__ = dict.__setitem__

    def to_ns(d):
        return NS({k: to_ns(v) if isinstance(v, dict) else v for k, v in d.items()})
    torch.set_num_threads(threads)
    steps = 100
    config = to_ns(yaml.safe_load(open(os.path.join(rh.REF, 'models/DDNM/configs/imagenet_256.yml'))))
    cfg = ounet.make_config(256, 256, 2, "32,16,8", 64, True)
    w = ounet.random_weights(cfg, 12)
    model = ref_unet(FULL, w)
    masked, masks, tape = ddnm_full_inputs(seed, n_img, steps)
    xs_all, out_all, ks = [], [], list(range(every - 1, steps, every))
    orig_to, orig_randn, orig_randn_like = torch.Tensor.to, torch.randn, torch.randn_like
    for im in range(n_img):
        args = NS(sigma_y=0, eta=0.85, seed=1234)
        runner = diffusion.Diffusion(args, config, device=torch.device('cpu'))
        pos = {'i': 0}
        seen = []
        t0 = time.time()

        def next_noise(*a, **k):
            n = torch.from_numpy(tape[im, pos['i']][None].copy())
            pos['i'] += 1
            return n

        def wrapped(x, t):
            k = len(seen)                                  # x = the state after update k - 1
            assert float(t[0]) == 990.0 - 10.0 * k
            seen.append(x.detach()[:, :, ::stride, ::stride].clone() if (k - 1) in ks else None)
            with torch.no_grad():
                y = model(x, t)
            if k % 10 == 0:
                print(name, 'image', im, 'step', k, 'x std %.4f' % float(x.std()), '%.0f s' % (time.time() - t0), flush=True)
            return y

        def to(self, *a, **k):
            if a and a[0] == 'cuda':
                return self
            return orig_to(self, *a, **k)
        last = []
        orig_inv = diffusion.inverse_data_transform

        def inv(config_, xi):                              # the un-clamped final state x_0 is what the sampler hands to this
            last.append(xi.detach().clone())
            return orig_inv(config_, xi)
        torch.Tensor.to, torch.randn, torch.randn_like = to, next_noise, next_noise
        diffusion.inverse_data_transform = inv
        try:
            out = runner.simplified_ddnm_inpainting(wrapped, torch.from_numpy(masked[im:im + 1]).unsqueeze(0),
                                                    torch.from_numpy(masks[im:im + 1]))
        finally:
            torch.Tensor.to, torch.randn, torch.randn_like = orig_to, orig_randn, orig_randn_like
            diffusion.inverse_data_transform = orig_inv
        assert pos['i'] == steps + 1 and len(seen) == steps and len(last) == 1
        out = out[0, 0].numpy() if out.dim() == 5 else out[0].numpy()
        assert out.shape == (3, 256, 256) and out.min() >= 0.0 and out.max() <= 1.0
        xs = [seen[k + 1][0].numpy() for k in ks[:-1]] + [last[0].reshape(3, 256, 256)[:, ::stride, ::stride].numpy()]
        xs_all.append(np.stack(xs))
        out_all.append(out.copy())
        np.savez_compressed(os.path.join(OUT, name), seed=seed, steps=steps, n_img=im + 1, stride=stride, weight_seed=12,
                            ks=np.array(ks), xs=np.stack(xs_all), out=np.stack(out_all))
        print(name, 'image', im, 'done in %.0f s' % (time.time() - t0), flush=True)


def main():
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['ddnm', 'small', 'full']
    if 'ddnm' in which:
        gen_ddnm('ddnm_sampler.npz')
    if 'small' in which:
        gen_unet('unet_small.npz', SMALL, seed=11, batch=2, stride=1)
    if 'full' in which:
        gen_unet('unet_full.npz', FULL, seed=12, batch=1, stride=8)
    if 'small16' in which:
        gen_unet_fp16('unet_small_fp16.npz', SMALL, seed=11, batch=2, stride=1, src='unet_small.npz', boosts=(4096.0, 6144.0))
    if 'full16' in which:
        gen_unet_fp16('unet_full_fp16.npz', FULL, seed=12, batch=1, stride=8, src='unet_full.npz', boosts=(2048.0, 4096.0))
    if 'ddnm_full' in which:
        gen_ddnm_full('ddnm_unet_full.npz')
    if 'ddnm_full16' in which:
        gen_ddnm_full('ddnm_unet_full_fp16.npz', fp16=True)
    if 'ddnm_full100' in which:
        gen_ddnm_full100('ddnm_unet_full100.npz')


if __name__ == '__main__':
    main()
