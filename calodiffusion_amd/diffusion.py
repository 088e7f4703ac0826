"""Generic diffusion wrapper: sampling, generation loop, loss (mirror of reference calodiffusion/models/diffusion.py)."""
from __future__ import annotations

import copy
from abc import ABC, abstractmethod
from typing import Callable, Optional, Union

import numpy as np
import torch

from . import utils
from .configs import load_config


class Diffusion(torch.nn.Module, ABC):
    """Same constructor and public methods as the reference class (models/diffusion.py:18-197)."""

    def __init__(self, config: Union[str, dict], n_steps: int = 400, loss_type: str = "l2"):
        super().__init__()
        self.config = load_config(config)
        self.device = utils.get_device()
        self.nsteps = n_steps
        self.loss_type = loss_type
        self.hgcal = self.pre_embed = False
        loss_algo = self.config.get("TRAINING_OBJ", "noise_pred")
        self.loss_function = utils.load_attr("loss", loss_algo)(self.config, self.nsteps, self.loss_type)
        sampler_algo = self.config.get("SAMPLER", "DDim")
        self.sampler_algorithm = utils.load_attr("sampler", sampler_algo)(self.config)
        self.NN_embed = None
        if "orig" not in self.config.get("SHOWER_EMBED", ""):
            self._data_shape = self.config["SHAPE_PAD"][1:]
        else:
            self._data_shape = self.config["SHAPE_ORIG"][1:]
        # device Philox stream for noise_generation: (seed, running offset).  A rank of a batch-sharded job calls
        # set_noise_shard(): every rank then walks the SAME global stream and draws only its rows of every tensor, so the
        # union of the shards is the single-GPU result of the same seed.
        self.noise_seed = int(self.config.get("SEED", 1234))
        self.noise_offset = 0
        self.noise_shard = None  # (first row of this rank, rows of the global batch)

    @abstractmethod
    def init_model(self):
        raise NotImplementedError

    def init_embedding_model(self):
        return None

    def set_noise_shard(self, first_row: int, global_batch: int):
        """This process samples rows [first_row, first_row + B) of a global batch of `global_batch` showers
        (utils.shard_batch); None-like (0, 0) switches sharding off."""
        self.noise_shard = (int(first_row), int(global_batch)) if global_batch else None

    def _shard_geometry(self, shape):
        """(elements per row, first element of this shard in a global tensor, elements of the global tensor)"""
        per = int(np.prod(shape[1:]))
        if self.noise_shard is None:
            return per, 0, int(shape[0]) * per
        lo, gb = self.noise_shard
        if lo + shape[0] > gb:
            raise ValueError(f"noise shard rows [{lo}, {lo + shape[0]}) exceed the global batch {gb}")
        return per, lo * per, gb * per

    @abstractmethod
    def noise_generation(self, shape):
        """Unit normal start tensor (diffusion.py:58-61); drawn from the device Philox stream (this rank's rows of it)."""
        from .engine import randn
        _, first, total = self._shard_geometry(shape)
        out = randn(shape, self.device, self.noise_seed, self.noise_offset + first)
        self.noise_offset += total
        return out

    def step_noise_stream(self, start):
        """(offset, stride) of a sampler's per-step noise tensors: they follow the start tensor in the stream, one global
        tensor apart (samplers call this after noise_generation has advanced the stream past the start tensor)."""
        _, first, total = self._shard_geometry(start.shape)
        return self.noise_offset + first, total

    @abstractmethod
    def forward(self):
        raise NotImplementedError

    @abstractmethod
    def __call__(self, x_noisy, E, sigma, model, layers):
        raise NotImplementedError

    def sample(self, energy: torch.Tensor, layers, num_steps: int = 400, debug: bool = False,
               sample_offset: Optional[int] = 0, start: Optional[torch.Tensor] = None):
        """diffusion.py:77-104.  ``start`` (optional, parity hook) replaces the internally drawn noise."""
        shape = [energy.shape[0]] + list(copy.copy(self._data_shape))
        if start is None:
            start = self.noise_generation(shape)
        x, xs, x0s = self.sampler_algorithm(self, start, energy, layers, num_steps, sample_offset, debug)
        # every noise tensor the sampler drew took one (global) tensor's worth of normals behind the start tensor's own
        self.noise_offset += self._shard_geometry(shape)[2] * getattr(self.sampler_algorithm, "noise_tensors_drawn", 0)
        if debug:
            return x.detach().cpu().numpy(), xs, x0s
        return x.detach().cpu().numpy()

    def compute_loss(self, data, energy, noise, layers, time=None, rnd_normal=None):
        """diffusion.py:106-110 (like the reference, `time` is not forwarded)."""
        return self.loss_function(self, data, energy, noise=noise, layers=layers, rnd_normal=rnd_normal)

    def generate(self, data_loader, sample_steps: int, debug: bool = False, sample_offset: Optional[int] = 0,
                 sparse_decoding: Optional[bool] = False, sparse_per_batch: Optional[bool] = False,
                 reverse_norm: Optional[Callable] = None):
        """Sampling loop over a loader of (E, layers, data) batches (diffusion.py:118-197).

        The inverse pre-processing (``utils.ReverseNorm``, diffusion.py:171-195) runs on the device for the regular-grid
        configs (``postprocess.ReverseNorm``: Dataset-2 / Dataset-3 shower maps; needs the EMAX / EMIN / logE / MAXDEP / ECUT
        keys of the reference's configs).  ``reverse_norm`` = a callable (generated, energies, layers, config) overrides it
        (e.g. the reference's own function for the geometry-converted datasets); ``reverse_norm=False`` returns the
        normalised-space showers.
        """
        self._physical_form(reverse_norm)  # raises NOW, not after minutes of sampling, if there is no inverse pre-processing
        generated, energies, layers = [], [], []
        for E, layers_, d_batch in data_loader:
            E = E.to(device=self.device)
            layers_ = layers_.to(device=self.device)
            out = self.sample(E, layers=layers_, num_steps=sample_steps, debug=debug, sample_offset=sample_offset)
            generated.append(out[0] if debug else out)
            energies.append(E.detach().cpu().numpy())
            if "layer" in self.config["SHOWERMAP"]:
                layers.append(layers_.detach().cpu().numpy())
        generated, energies = np.concatenate(generated), np.concatenate(energies)
        layers = np.concatenate(layers) if layers else None
        return self._to_physical(generated, energies, layers, reverse_norm, debug)

    def _physical_form(self, reverse_norm) -> str:
        """Which inverse pre-processing generate() will apply: 'callable', 'device', or 'none' (reverse_norm=False).  Raises for
        configs the device form does not cover -- called at the top of generate() (and of LayerDiffusion.generate), before the
        sampling loop."""
        cfg = self.config
        if callable(reverse_norm):
            return "callable"
        if reverse_norm is False:
            return "none"
        if (reverse_norm is None and cfg.get("DATASET_NUM", 2) in (2, 3)
                and cfg.get("SHOWERMAP") in ("layer-logit-norm", "logit-norm")
                and all(k in cfg for k in ("EMAX", "EMIN", "logE", "MAXDEP", "ECUT"))):
            return "device"
        # the reference always applies utils.ReverseNorm (diffusion.py:171-195): never hand back normalised-space showers
        # silently.  HGCal / Dataset-1 need geometry files outside this package: pass the reference's function.
        raise ValueError(
            "generate(): no inverse pre-processing for this config on the device path (needs DATASET_NUM 2/3 with a "
            "[layer-]logit-norm SHOWERMAP and the EMAX/EMIN/logE/MAXDEP/ECUT keys); pass reverse_norm=<callable "
            "(generated, energies, layers, config)> or reverse_norm=False for normalised-space showers")

    def _to_physical(self, generated, energies, layers, reverse_norm, debug=False):
        """Inverse pre-processing of generated showers (shared with LayerDiffusion.generate)."""
        cfg = self.config
        device_form = self._physical_form(reverse_norm) == "device"
        if callable(reverse_norm):
            generated, energies = reverse_norm(generated, energies, layers, cfg)
        elif device_form:
            from .postprocess import ReverseNorm
            generated, energies = ReverseNorm(generated, energies, shape=cfg["SHAPE_FINAL"], config=cfg, emax=cfg["EMAX"],
                                              emin=cfg["EMIN"], layerE=layers, logE=cfg["logE"], max_deposit=cfg["MAXDEP"],
                                              showerMap=cfg["SHOWERMAP"], dataset_num=cfg.get("DATASET_NUM", 2),
                                              ecut=float(cfg["ECUT"]))
            generated = generated.reshape(cfg["SHAPE_ORIG"])
        return generated, np.reshape(energies, (energies.shape[0], -1))
