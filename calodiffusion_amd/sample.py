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
