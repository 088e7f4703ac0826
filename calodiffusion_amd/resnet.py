"""Parameter container for the conditional residual MLP that LayerDiffusion samples layer energies with.

Mirrors the interface of the reference ``ResNet`` / ``ResDense`` (calodiffusion/models/models.py:373-457): same constructor
arguments, the same ``forward(x, cond, time)`` signature and the same ``state_dict`` keys, shapes and order
(``time_mlp.{1,3,5}``, ``cond_mlp.{0,2,4}``, ``in_lay``, ``hidden_layers.i.{embeder.1, dense1.0, dense2.0}``, ``out_lay``).
Sub-modules are created in the reference's order with the stock initialisers, including the one ``nn.Linear`` the reference
builds and discards (models.py:429), so that ``torch.manual_seed(s)`` + construction gives bit-identical parameters.

Storage only: ``forward`` hands device pointers to the HIP library (``cd_layer_forward``).
"""
from __future__ import annotations

import torch.nn as nn

from .unet import _Holder


class _Slot(_Holder):
    """Occupies an index of an nn.Sequential that the reference fills with a parameter-free module (GELU / Unflatten)."""


class ResDense(_Holder):
    def __init__(self, dim, dim_out, cond_emb_dim=128):
        super().__init__()
        self.embeder = nn.Sequential(_Slot(), nn.Linear(cond_emb_dim, dim_out))
        self.dense1 = nn.Sequential(nn.Linear(dim, dim_out), _Slot())
        self.dense2 = nn.Sequential(nn.Linear(dim_out, dim_out), _Slot())


class ResNet(nn.Module):
    def __init__(self, dim_in=45, num_layers=3, hidden_dim=256, cond_emb_dim=128, cond_size=1):
        super().__init__()
        half = cond_emb_dim // 2
        time_layers = [_Slot(), nn.Linear(1, half // 2), _Slot(), nn.Linear(half // 2, half), _Slot(), nn.Linear(half, half)]
        cond_layers = [nn.Linear(cond_size, half // 2), _Slot(), nn.Linear(half // 2, half), _Slot(), nn.Linear(half, half)]
        self.time_mlp = nn.Sequential(*time_layers)
        self.cond_mlp = nn.Sequential(*cond_layers)
        nn.Linear(dim_in + cond_emb_dim, dim_in)  # models.py:429 builds and drops this layer: keep the RNG stream aligned
        self.in_lay = nn.Linear(dim_in, hidden_dim)
        self.hidden_layers = nn.ModuleList([ResDense(hidden_dim, hidden_dim, cond_emb_dim=cond_emb_dim)
                                            for _ in range(num_layers - 1)])
        self.out_lay = nn.Linear(hidden_dim, dim_in)
        self.dim_in, self.hidden_dim, self.cond_emb_dim, self.cond_size = dim_in, hidden_dim, cond_emb_dim, cond_size
        self._engine = None
        self._engine_opts = {}

    def engine(self):
        """The HIP binding of this parameter set (created on first use; needs a GPU)."""
        if self._engine is None:
            from .engine import LayerMlpEngine
            self._engine = LayerMlpEngine(self, **self._engine_opts)
        return self._engine

    def forward(self, x, cond=None, time=None, controls=None):
        """ResNet.forward (models.py:444-457): x (B, dim_in), cond (B, cond_size), time (B) -> (B, dim_in)."""
        return self.engine().forward(x, cond, time)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = None
        return out
