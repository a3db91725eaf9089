"""Row D1: the DDNM inpainting sampler, restated in torch fp32 on the CPU.  (oracle -- test infrastructure)

Follows /root/reference/models/DDNM/guided_diffusion/diffusion.py:46-76 (get_beta_schedule 'linear'),
:79-113 (Diffusion.__init__), :809-812 (compute_alpha), :770-791 (get_schedule_jump),
:459-570 (simplified_ddnm_inpainting, sigma_y = 0, eta = 0.85), datasets/__init__.py:208-235
(data_transform x -> 2x-1, inverse_data_transform clamp((x+1)/2, 0, 1)) and
/root/reference/models/DDNM/ddnm_inpainting.py:15-44 (Inpainter) with configs/imagenet_256.yml
(T_sampling 100, travel_length 1, travel_repeat 1, 1000 diffusion steps, beta 1e-4..0.02).
Noise is INJECTED (x_T and one epsilon per step): torch RNG streams do not reproduce across devices.
"""
import numpy as np
import torch

ETA = 0.85
NUM_DIFFUSION_TIMESTEPS = 1000
T_SAMPLING = 100


def betas_linear(beta_start=1e-4, beta_end=0.02, n=NUM_DIFFUSION_TIMESTEPS):
    return torch.from_numpy(np.linspace(beta_start, beta_end, n, dtype=np.float64)).float()


def compute_alpha(betas, t):
    """diffusion.py:809-812: alpha_bar(t) = cumprod(1 - [0, beta])[t + 1] in float32 (so alpha_bar(-1) = 1)."""
    b = torch.cat([torch.zeros(1), betas], dim=0)
    return (1 - b).cumprod(dim=0)[t + 1]


def time_pairs(T_sampling=T_SAMPLING, n_steps=NUM_DIFFUSION_TIMESTEPS):
    """diffusion.py:504-519 with travel_length = travel_repeat = 1: (t, t_next) = (990, 980) ... (0, -1)."""
    skip = n_steps // T_sampling
    times = list(range(T_sampling - 1, -1, -1)) + [-1]
    pairs = []
    for i, j in zip(times[:-1], times[1:]):
        i, j = i * skip, j * skip
        if j < 0:
            j = -1
        pairs.append((i, j))
    return pairs


def step_coefficients(betas=None, eta=ETA):
    """Per step float32 scalars (a_t, a_next, c0..): exactly the tensors the reference forms per iteration."""
    betas = betas_linear() if betas is None else betas
    out = []
    for t, tn in time_pairs():
        at = compute_alpha(betas, torch.tensor(t))
        an = compute_alpha(betas, torch.tensor(tn))
        sigma_t = (1 - an ** 2).sqrt()
        c1 = (1 - an).sqrt() * eta
        c2 = (1 - an).sqrt() * ((1 - eta ** 2) ** 0.5)
        out.append(dict(t=t, t_next=tn, at=at, at_next=an, sqrt_1m_at=(1 - at).sqrt(), sqrt_at=at.sqrt(),
                        sqrt_at_next=an.sqrt(), sigma_t=sigma_t, c1=c1, c2=c2))
    return out


def ddnm_step(xt, et, y, mask, co, eps):
    """One iteration of diffusion.py:529-552 (sigma_y = 0 => lambda_t = 1, gamma_t = sigma_t).
    xt, et, y, eps [N,3,H,W]; mask [N,1,H,W] or broadcastable."""
    x0_t = (xt - et * co['sqrt_1m_at']) / co['sqrt_at']
    x0_hat = x0_t - 1.0 * (mask * ((mask * x0_t) - y))
    return co['sqrt_at_next'] * x0_hat + co['sigma_t'] * (co['c1'] * eps + co['c2'] * et)


def sample(model_fn, masked_img, mask, x_T, eps_list, n_steps=None):
    """simplified_ddnm_inpainting for one image batch.  masked_img [N,3,H,W] in [0,1], mask [N,H,W] (1 = keep),
    x_T [N,3,H,W], eps_list[k] [N,3,H,W].  model_fn(x, t[N]) -> [N,6 or 3,H,W].  Returns [N,3,H,W] in [0,1]."""
    cos = step_coefficients()
    if n_steps is not None:
        cos = cos[:n_steps]
    m = mask[:, None].float()
    x_orig = 2 * masked_img.float() - 1.0
    y = x_orig * m
    x = x_T.float()
    for k, co in enumerate(cos):
        t = torch.ones(x.shape[0]) * co['t']
        et = model_fn(x, t)
        if et.shape[1] == 6:
            et = et[:, :3]
        x = ddnm_step(x, et, y, m, co, eps_list[k])
    return torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)
