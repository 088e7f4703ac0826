"""CaloDiffusion: the EDM-preconditioned cylindrical U-Net denoiser (mirror of reference
calodiffusion/models/calodiffusion.py:9-173) running on the HIP engine."""
from __future__ import annotations

from typing import Union

import torch

from . import utils
from .diffusion import Diffusion
from .unet import CondUnet, unet_kwargs_from_config


class CaloDiffusion(Diffusion):
    def __init__(self, config: Union[str, dict], n_steps: int = 400, loss_type: str = "l2"):
        super().__init__(config, n_steps, loss_type)
        config = self.config
        self.pre_embed = "pre-embed" in config.get("SHOWER_EMBED", "")
        self.hgcal = config.get("HGCAL", False)
        self.fully_connected = "FCN" in config.get("SHOWER_EMBED", "")
        self.time_embed = config.get("TIME_EMBED", "sin")
        self.dataset_num = config.get("DATASET_NUM", 2)
        self.training_objective = config.get("TRAINING_OBJ", "noise_pred")
        self.layer_cond = "layer" in config.get("SHOWERMAP", "")
        if self.fully_connected:
            raise NotImplementedError("the FCN/ResNet layer model is outside the HIP hot path (SURVEY.md 8f rank 3)")
        if "NN" in config.get("SHOWER_EMBED", "") and not self.pre_embed:
            raise NotImplementedError("in-model geometry embeddings (NNConverter / HGCalConverter) need binning files that "
                                      "are not part of the hot path; use a pre-embedded ('...-pre-embed') dataset")
        if self.time_embed not in ("log", "sigma"):
            raise KeyError(self.time_embed)  # the reference's do_time_embed raises the same way (calodiffusion.py:148-152)
        self.model = self.init_model()
        self.NN_embed = None
        self.do_embed = False

    # ------------------------------------------------------------------ construction / weights
    def init_model(self):
        cfg = self.config
        unet = CondUnet(**unet_kwargs_from_config(cfg))
        objective = type(self.loss_function).__name__
        if "noise_pred" in objective:
            obj = "noise_pred"
        elif "mean_pred" in objective:
            obj = "mean_pred"
        elif "hybrid" in objective:
            obj = "hybrid"
        else:
            raise ValueError("??? Training obj %s" % objective)
        unet._engine_opts = dict(
            rz_input=cfg.get("R_Z_INPUT", False), phi_input=cfg.get("PHI_INPUT", False), time_kind=self.time_embed,
            objective=obj, sigma_data=self.loss_function.sigma_data,
            coords=utils.coordinate_profiles(self.dataset_num, cfg["SHAPE_FINAL"][2:]))
        return unet.to(self.device)

    def load_state_dict(self, state_dict, strict=True):
        """Prefix-tolerant loading, as the reference (calodiffusion.py:31-37)."""
        base = list(state_dict.keys())[10].split(".")[0]
        if base != "model":
            state_dict = {k.removeprefix(f"{base}."): v for k, v in state_dict.items() if k.split(".")[0] == base}
        return super().load_state_dict(state_dict, strict)

    def engine(self):
        return self.model.engine()

    def to(self, *a, **k):
        out = super().to(*a, **k)
        self.device = next(self.parameters()).device
        return out

    # ------------------------------------------------------------------ hot path
    def noise_generation(self, shape):
        return super().noise_generation(shape)

    def cond_tensor(self, E, layers):
        """cat(E, layers) when the config conditions on layer energies (calodiffusion.py:89-90)."""
        if self.layer_cond and layers is not None:
            E = torch.cat([E, layers], dim=1)
        return E.to(torch.float32).contiguous()

    def forward(self, x, E, time, layers, controls=None):
        """calodiffusion.py:86-98: x is the already c_in-scaled input; returns the raw network output F."""
        if controls is not None:
            raise NotImplementedError("ControlNet is dead code in the reference")
        rz_phi = self.add_RZPhi(x).float()
        return self.model(rz_phi, cond=self.cond_tensor(E, layers), time=time.float())

    def add_RZPhi(self, x):
        """calodiffusion.py:121-142 (only used by the generic `forward`; `denoise` synthesises the channels in-kernel)."""
        cats = [x]
        shape = (x.shape[0], 1) + tuple(x.shape[2:])
        r, z, phi = utils.coordinate_profiles(self.dataset_num, x.shape[2:])
        if self.config.get("R_Z_INPUT", False):
            cats.append(torch.from_numpy(r).to(x.device).view(1, 1, 1, 1, -1).expand(shape))
            cats.append(torch.from_numpy(z).to(x.device).view(1, 1, -1, 1, 1).expand(shape))
        if self.config.get("PHI_INPUT", False):
            cats.append(torch.from_numpy(phi).to(x.device).view(1, 1, 1, -1, 1).expand(shape))
        return torch.cat(cats, dim=1) if len(cats) > 1 else x

    def do_time_embed(self, sigma=None):
        embed = {"sigma": lambda s: s / (1 + s ** 2).sqrt(), "log": lambda s: 0.5 * torch.log(s)}
        return embed[self.time_embed](sigma)

    def denoise(self, x, E=None, sigma=None, layers=None, controls=None):
        """EDM-preconditioned denoiser (calodiffusion.py:154-169): one C-ABI call."""
        if controls is not None:
            raise NotImplementedError("ControlNet is dead code in the reference")
        return self.engine().denoise(x, sigma.reshape(-1), self.cond_tensor(E, layers))

    def __call__(self, x, **kwargs):
        return self.denoise(x, **kwargs)
