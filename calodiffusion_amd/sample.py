"""Samplers of the hot path: DDim and DDPM (mirror of reference calodiffusion/models/sample.py:17-121).

The reference's loop does, per step, five host<->device `extract` round trips, ~12 elementwise launches, one unused
`randn` and a callback into the model.  Here the whole loop is one C-ABI call: the per-step scalars are tabulated on the
host once, one step is captured as a hipGraph and replayed.
"""
from __future__ import annotations

from typing import Any

import torch

from . import schedule


class Sample:
    def __init__(self, config) -> None:
        self.config = config
        self.sample_config = self.config.get("SAMPLER_OPTIONS", {})

    def __call__(self, model, start, energy, layers, num_steps, sample_offset, debug) -> Any:
        raise NotImplementedError


class DDim(Sample):
    """Deterministic sampler (eta = 0).  Returns (x, xs, x0s); the trajectories are only recorded when ``debug``
    (the reference keeps 2N full tensors alive regardless and drops them in Diffusion.sample, diffusion.py:91,104)."""

    def __init__(self, config):
        super().__init__(config)
        self.ddim_eta = 0.0
        self.use_graph = bool(self.sample_config.get("HIP_GRAPH", True))
        self.seed = int(self.sample_config.get("SEED", 0))
        self.step_noise = None  # parity hook: (n_steps, B, 1, D, H, W) noise to use instead of the device Philox stream

    @torch.no_grad()
    def __call__(self, model, start, energy, layers, num_steps, sample_offset=0, debug=False) -> Any:
        table = schedule.ddim_step_table(num_steps, self.ddim_eta, sample_offset or 0)
        x, xs, x0s = model.engine().ddim_sample(
            start, model.cond_tensor(energy, layers), table, step_noise=self.step_noise, seed=self.seed,
            offset=getattr(model, "noise_offset", 0) + start.numel(), debug=debug, use_graph=self.use_graph)
        if debug:
            return x, list(xs.unbind(0)), list(x0s.unbind(0))
        return x, [], []


class DDPM(DDim):
    """Stochastic version (eta = 1), models/sample.py:112-121."""

    def __init__(self, config):
        super().__init__(config)
        self.ddim_eta = 1.0


class Euler(Sample):
    """EDM first-order (Euler) sampler on the Karras noise schedule (reference models/sample.py:577-727, 771-789), deterministic
    form: NOISY_SAMPLE off (S_churn = 0), which is the configuration default.  The update x + (t_next - t)(x - x0)/t equals
    x0 + t_next (x - x0)/t, i.e. the device loop of DDim with the step table (t_i, t_{i+1}, 0, 1): the whole trajectory is one
    C-ABI call with one captured step graph, like DDim.

    SAMPLER_OPTIONS: RHO, SIGMA_MIN, SIGMA_MAX (defaults 7, 0.002, 80).  NOISY_SAMPLE / ORG_SCHEDULE are not provided."""

    def __init__(self, config):
        super().__init__(config)
        if self.config.get("NOISY_SAMPLE", False):
            raise NotImplementedError("Euler: the stochastic (NOISY_SAMPLE / S_churn > 0) variant is not provided")
        if self.sample_config.get("ORG_SCHEDULE", False):
            raise NotImplementedError("Euler: ORG_SCHEDULE (iDDPM time steps) is not provided")
        self.sigma_min = self.sample_config.get("SIGMA_MIN", 0.002)
        self.sigma_max = self.sample_config.get("SIGMA_MAX", 80.0)
        self.rho = self.sample_config.get("RHO", 7)
        self.use_graph = bool(self.sample_config.get("HIP_GRAPH", True))

    @torch.no_grad()
    def __call__(self, model, start, energy, layers, num_steps, sample_offset=0, debug=False) -> Any:
        table = schedule.edm_euler_step_table(num_steps, sample_offset or 0, sigma_min=self.sigma_min, sigma_max=self.sigma_max,
                                              rho=self.rho)
        x, xs, x0s = model.engine().ddim_sample(start, model.cond_tensor(energy, layers), table, debug=debug,
                                                use_graph=self.use_graph)
        if debug:
            return x, list(xs.unbind(0)), list(x0s.unbind(0))
        return x, [], []
