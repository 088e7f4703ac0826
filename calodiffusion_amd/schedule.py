"""Noise schedules and per-step sampler scalars, computed on the host in fp32 in the reference's op order
(utils/sampling.py:16-24, models/sample.py:45-101), so that the step table handed to the device is bit-identical
to what the reference's loop would gather each iteration (five `extract` device syncs per step there)."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch


def cosine_beta_schedule(nsteps: int, s: float = 0.008) -> torch.Tensor:
    x = torch.linspace(0, nsteps, nsteps + 1)
    alphas_cumprod = torch.cos(((x / nsteps) + s) / (1 + s) * np.pi * 0.5) ** 2
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0.0001, 0.9999)


def tables(nsteps: int) -> Dict[str, torch.Tensor]:
    betas = cosine_beta_schedule(nsteps)
    ac = torch.cumprod(1.0 - betas, axis=0)
    ac_prev = torch.nn.functional.pad(ac[:-1], (1, 0), value=1.0)
    return {"betas": betas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev,
            "sqrt_alphas_cumprod": torch.sqrt(ac), "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac)}


def ddim_step_table(num_steps: int, eta: float, sample_offset: int = 0) -> np.ndarray:
    """(n, 4) fp32 rows (sigma, sigma_prev*[t>0], ddim_sigma, denom) for t = num_steps-1-sample_offset .. 0."""
    tb = tables(num_steps)
    ac, ac_prev = tb["alphas_cumprod"], tb["alphas_cumprod_prev"]
    sa, s1 = tb["sqrt_alphas_cumprod"], tb["sqrt_one_minus_alphas_cumprod"]
    t = torch.flip(torch.arange(num_steps), [0])
    if sample_offset > 0:
        t = t[sample_offset:]
    sigma = s1[t] / sa[t]
    alpha, alpha_prev = ac[t], ac_prev[t]
    denom = sa[torch.clamp(t - 1, min=0)]
    ddim_sigma = eta * (((1 - alpha_prev) / (1 - alpha)) * (1 - alpha / alpha_prev)) ** 0.5
    sigma_prev = (1.0 - alpha_prev - ddim_sigma ** 2).sqrt() / denom
    sigma_prev = (t > 0).to(torch.float32) * sigma_prev
    return torch.stack([sigma, sigma_prev, ddim_sigma * torch.ones_like(sigma), denom], dim=1).numpy().astype(np.float32)


def edm_time_steps(num_steps: int, sample_offset: int = 0, sigma_min: float = 0.002, sigma_max: float = 80.0,
                   rho: float = 7) -> torch.Tensor:
    """EDM (Karras et al.) noise levels t_0 > ... > t_{N-1} > t_N = 0 in the reference's fp32 operation order
    (EDMAbstract.setup, models/sample.py:688-702; the iDDPM 'ORG_SCHEDULE' variant is not provided)."""
    step_indices = torch.arange(num_steps, dtype=torch.float32)
    t_steps = (sigma_max ** (1 / rho) + step_indices / (num_steps - 1)
               * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    t_steps = torch.cat([torch.as_tensor(t_steps), torch.zeros_like(t_steps[:1])])
    return t_steps[sample_offset:]


def edm_euler_step_table(num_steps: int, sample_offset: int = 0, **kw) -> np.ndarray:
    """Rows (sigma, sigma_prev, 0, 1) that turn the device sampler loop (x <- x0 + sigma_prev * (x - x0) / sigma) into the
    deterministic EDM Euler sampler x <- x + (t_next - t) * (x - x0) / t (models/sample.py:771-789 with S_churn = 0)."""
    t = edm_time_steps(num_steps, sample_offset, **kw)
    n = t.numel() - 1
    return torch.stack([t[:-1], t[1:], torch.zeros(n), torch.ones(n)], dim=1).numpy().astype(np.float32)


def karras_steps(num_step: int, min_t, max_t, rho: float = 7) -> torch.Tensor:
    """num_step Karras noise levels from max_t down to min_t in fp32 (utils/sampling.py:44-51; the Restart sampler's
    excursions)."""
    step_indices = torch.arange(num_step, dtype=torch.float32)
    return (max_t ** (1 / rho) + step_indices / (num_step - 1) * (min_t ** (1 / rho) - max_t ** (1 / rho))) ** rho
