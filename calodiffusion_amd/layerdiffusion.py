"""LayerDiffusion: two-stage generation -- a small MLP diffusion model draws the (B, D+1) {total, per-layer} energies, the
U-Net then generates the shower conditioned on them (mirror of reference calodiffusion/models/layerdiffusion.py:12-235).

Both stages run on the HIP library: the layer stage is ONE launch for the whole trajectory (``cd_layer_sample``), the shower
stage is the U-Net sampler loop of ``CaloDiffusion``.
"""
from __future__ import annotations

import copy
import os
from typing import Optional

import numpy as np
import torch

from . import utils
from .calodiffusion import CaloDiffusion
from .resnet import ResNet


class LayerDiffusion(CaloDiffusion):
    def __init__(self, config, n_steps=400, loss_type="l2"):
        super().__init__(config, n_steps, loss_type)
        self.layer_loss = False
        sampler_algo = self.config.get("LAYER_SAMPLER", "DDim")
        self.layer_sampler = utils.load_attr("sampler", sampler_algo)(self.config)
        self.layer_steps = self.config.get("LAYER_STEPS", n_steps)
        self.shape_pad = self.config.get("SHAPE_PAD")
        if self.shape_pad is None:
            self.shape_pad = self.config["SHAPE_FINAL"]

    # ------------------------------------------------------------------ construction / weights
    def init_model(self):
        """layerdiffusion.py:35-40: the layer model is built first (its parameters come first in the RNG stream)."""
        cond_size = 3 if self.hgcal else 1
        self.layer_model = ResNet(dim_in=self.config["SHAPE_FINAL"][2] + 1, num_layers=5, cond_size=cond_size)
        model = super().init_model()
        self.layer_model._engine_opts = dict(time_kind=self.time_embed, objective=model._engine_opts["objective"],
                                             sigma_data=self.loss_function.sigma_data)
        self.layer_model.to(self.device)
        self.base_model = model
        return model

    def set_layer_state(self, is_layer=False):
        """layerdiffusion.py:42-50."""
        self.layer_loss = bool(is_layer)
        self.model = self.layer_model if is_layer else self.base_model

    def engine(self):
        return self.model.engine()

    def cond_tensor(self, E, layers):
        if self.layer_loss:  # layer_forward conditions on the incident energy only (layerdiffusion.py:109-112)
            return E.to(torch.float32).contiguous()
        return super().cond_tensor(E, layers)

    def forward(self, x, E, time, layers=None, controls=None, **kwargs):
        if self.layer_loss:
            return self.layer_forward(x, E, time)
        return super().forward(x, E, time, layers, controls)

    def layer_forward(self, x, E, time, **kwargs):
        """layerdiffusion.py:109-112 (add_RZPhi leaves a 2-D tensor unchanged)."""
        return self.layer_model(x, cond=E.to(torch.float32), time=time.to(torch.float32))

    def compute_loss(self, data, energy, noise, layers, time=None, rnd_normal=None):
        """layerdiffusion.py:52-57: in the layer state the layer model is trained on the layer energies themselves (fresh noise
        of their shape); one cd_layer_train_step call gives the loss and every gradient."""
        if self.layer_loss:
            layers = layers.to(torch.float32)
            noise = self.noise_generation(layers.shape).to(torch.float32)
            return self.loss_function(self, layers, energy, noise=noise, layers=layers, rnd_normal=rnd_normal)
        return super().compute_loss(data, energy, noise, layers, time, rnd_normal)

    @staticmethod
    def _without_prefix(sd, prefixes):
        """Checkpoint keys with the first matching module prefix stripped (other modules' entries dropped)."""
        heads = {k.split(".", 1)[0] for k in sd}
        for pre in prefixes:
            if pre in heads:
                cut = len(pre) + 1
                return {k[cut:]: v for k, v in sd.items() if k.startswith(pre + ".")}
        return sd

    @staticmethod
    def _load_tolerant(module, sd, strict):
        """Strict load; shape or missing-key errors propagate, anything else (unexpected extras) retries non-strictly --
        the behaviour of both loaders of the reference (layerdiffusion.py:76-85, 97-105)."""
        try:
            return module.load_state_dict(sd, strict=strict)
        except RuntimeError as err:
            if any(tag in str(err) for tag in ("size mismatch", "Missing key(s) in state_dict")):
                raise
            return module.load_state_dict(sd, strict=False)

    def load_layer_model_state(self, strict=True):
        """The layer model comes from its own checkpoint: config['layer_model'], else <checkpoint>/checkpoint.pth
        (layerdiffusion.py:59-85)."""
        path = self.config.get("layer_model")
        if path is None:
            path = os.path.join(self.config.get("checkpoint", ""), "checkpoint.pth")
            if not os.path.exists(path):
                raise RuntimeError("Could not load layer model from either config or checkpoint path")
        blob = torch.load(path, map_location=self.device, weights_only=False)
        sd = blob.get("model_state_dict", blob) if isinstance(blob, dict) else blob
        self._load_tolerant(self.layer_model, self._without_prefix(sd, ("layer_model",)), strict)

    def load_state_dict(self, state_dict, strict=True):
        """layerdiffusion.py:87-105: layer model from its own checkpoint, then the base U-Net from `state_dict`."""
        self.load_layer_model_state(strict)
        return self._load_tolerant(self.base_model, self._without_prefix(state_dict, ("base_model", "model")), strict)

    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        sd["layer_model"] = self.layer_model.state_dict()
        return sd

    # ------------------------------------------------------------------ hot path
    def denoise(self, x, E=None, sigma=None, layers=None, controls=None):
        if self.layer_loss:
            return self.layer_model.engine().denoise(x, sigma.reshape(-1), self.cond_tensor(E, None))
        return super().denoise(x, E=E, sigma=sigma, layers=layers, controls=controls)

    def sample_layers(self, energy, layers=None, debug=False, sample_offset=None, start: Optional[torch.Tensor] = None):
        """layerdiffusion.py:114-132: the (B, D+1) layer energies, one launch for the whole trajectory."""
        self.set_layer_state(is_layer=True)
        try:
            if start is None:
                start = self.noise_generation((energy.shape[0], self.shape_pad[2] + 1)).to(torch.float32)
            x, _, _ = self.layer_sampler(self, start, energy, layers, self.layer_steps, sample_offset, debug)
            self.noise_offset += start.numel() * self.layer_steps
        finally:
            self.set_layer_state(is_layer=False)
        return x

    def sample(self, energy: torch.Tensor, layers=None, num_steps: int = 400, debug: bool = False,
               sample_offset: Optional[int] = None, return_layers: bool = False, start: Optional[torch.Tensor] = None,
               layer_start: Optional[torch.Tensor] = None) -> dict:
        """layerdiffusion.py:134-169; returns {'x', ['xs', 'x0s'], ['layers']}.  ``start`` / ``layer_start`` (parity hooks)
        replace the internally drawn noise of the shower / layer stage."""
        shape = [energy.shape[0]] + list(copy.copy(self._data_shape))
        if start is None:
            start = self.noise_generation(shape).to(torch.float32)
        layers = self.sample_layers(energy, layers=None, debug=debug, sample_offset=sample_offset, start=layer_start)
        x, xs, x0s = self.sampler_algorithm(self, start, energy, layers, num_steps, sample_offset, debug)
        self.noise_offset += start.numel() * num_steps
        out = {"x": x.detach().cpu().numpy()}
        if debug:
            out["xs"], out["x0s"] = xs, x0s
        if return_layers:
            out["layers"] = layers
        return out

    def generate(self, data_loader, sample_steps: int, debug: bool = False, sample_offset: Optional[int] = 0,
                 sparse_decoding: Optional[bool] = False, sparse_per_batch: Optional[bool] = False, reverse_norm=None):
        """layerdiffusion.py:171-235: no layer energies are taken from the loader, the layer model generates them."""
        self._physical_form(reverse_norm)  # raise before sampling if this config has no inverse pre-processing here
        generated, energies, layers = [], [], []
        for E, _, _d in data_loader:
            E = E.to(device=self.device)
            out = self.sample(E, layers=None, num_steps=sample_steps, debug=debug, sample_offset=sample_offset,
                              return_layers=True)
            generated.append(out["x"])
            layers.append(out["layers"].detach().cpu().numpy())
            energies.append(E.detach().cpu().numpy())
        generated, energies, layers = np.concatenate(generated), np.concatenate(energies), np.concatenate(layers)
        return self._to_physical(generated, energies, layers, reverse_norm)
