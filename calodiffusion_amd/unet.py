"""Parameter container for the conditional cylindrical U-Net.

Mirrors the *interface* of the reference ``CondUnet`` (calodiffusion/models/models.py:523-748):
same constructor arguments, same ``forward(x, cond, time)`` signature, and -- so that
``torch.save``/``torch.load`` checkpoints interchange both ways -- the same ``state_dict`` key
names, shapes and ordering.  Sub-modules are created in the reference's construction order with
the stock torch initialisers, hence ``torch.manual_seed(s)`` followed by construction yields
bit-identical parameters (asserted by ``oracle/gen_golden.py`` and checked through committed
checksums in the tests).

The torch modules in here are *storage only*: none of their ``forward`` methods is ever called.
``CondUnet.forward`` hands raw device pointers to the HIP library (``calodiffusion_amd.engine``).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn


class _Holder(nn.Module):
    """A module that only owns parameters; computing with it directly is a bug."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("calodiffusion_amd parameter holders are storage only; use CondUnet.forward (HIP path)")


class _CylConv(_Holder):
    """Holds the Conv3d of a phi-periodic convolution under ``.conv`` (models.py:80-88)."""

    def __init__(self, cin, cout, kernel, stride=1, bias=True, pad_zr=0):
        super().__init__()
        pad = [pad_zr, 0, pad_zr]
        self.conv = nn.Conv3d(cin, cout, kernel_size=kernel, stride=stride, padding=pad, bias=bias)


class _CylConvT(_Holder):
    """Holds the ConvTranspose3d of the phi-periodic up-sampling under ``.convTrans`` (models.py:46-53)."""

    def __init__(self, c, kernel, stride, output_padding):
        super().__init__()
        self.convTrans = nn.ConvTranspose3d(c, c, kernel_size=kernel, stride=stride,
                                            padding=[1, kernel[1] - 1, 1], output_padding=output_padding)


class _Block(_Holder):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.proj = _CylConv(cin, cout, 3, pad_zr=1)
        try:
            self.norm = nn.GroupNorm(groups, cout)
        except ValueError:
            raise ValueError(f"Failed it init groupnorm with {groups} groups and {cout} out dims")
        self.act = nn.SiLU()


class _Resnet(_Holder):
    def __init__(self, cin, cout, cond_emb_dim, groups):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(cond_emb_dim, cout)) if cond_emb_dim is not None else None
        # the reference always instantiates the 1x1 projection (and so consumes RNG), then drops it when cin == cout
        proj = _CylConv(cin, cout, 1)
        self.block1 = _Block(cin, cout, groups)
        self.block2 = _Block(cout, cout, groups)
        self.res_conv = proj if cin != cout else nn.Identity()


class _LinAttn(_Holder):
    def __init__(self, dim, heads=1, dim_head=32):
        super().__init__()
        hidden = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.to_qkv = _CylConv(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(_CylConv(hidden, dim, 1), nn.GroupNorm(1, dim))


class _PreNorm(_Holder):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = nn.GroupNorm(1, dim)


class _Residual(_Holder):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


def _attn(dim):
    return _Residual(_PreNorm(dim, _LinAttn(dim)))


class CondUnet(nn.Module):
    """Drop-in for reference ``CondUnet`` (models.py:523-748) backed by the HIP engine.

    Supported: what the shipped configs instantiate -- cylindrical convolutions, ResNet blocks, linear attention, Linear
    cond/time embeddings -- plus the sinusoidal embeddings (``time_embed`` / ``cond_embed`` = True; with ``cond_embed`` the
    condition is one scalar per sample, ``cond`` of shape (B,), as the reference's broadcasting requires).  Anything else
    raises here, at construction, rather than silently computing something different.
    """

    def __init__(self, out_dim=1, layer_sizes=None, channels=1, cond_dim=128, resnet_block_groups=8,
                 use_convnext=False, mid_attn=False, block_attn=False, compress_Z=False, convnext_mult=2,
                 cylindrical=False, data_shape=(-1, 1, 45, 16, 9), time_embed=True, cond_embed=True,
                 cond_size=1, no_time=False):
        super().__init__()
        if use_convnext or not cylindrical or no_time or out_dim != 1:
            raise NotImplementedError(
                "HIP CondUnet supports the shipped configurations only: cylindrical ResNet U-Net with time conditioning, "
                "out_dim=1")
        layer_sizes = list(layer_sizes)
        self.channels, self.cond_dim, self.cond_size = channels, cond_dim, cond_size
        self.layer_sizes, self.groups = layer_sizes, resnet_block_groups
        self.block_attn, self.compress_Z = block_attn, compress_Z
        self.grid = tuple(int(v) for v in data_shape[-3:])
        half = cond_dim // 2
        g = resnet_block_groups
        zs = 2 if compress_Z else 1

        self.time_embed, self.cond_embed = bool(time_embed), bool(cond_embed)
        self.init_conv = _CylConv(channels, layer_sizes[0], 3, pad_zr=1)
        # (models.py:575-608) sinusoidal branch: SinusoidalPositionEmbeddings(half // 2) -- parameter-free, computed in the
        # embedding kernel -- stands where the first Linear + GELU of the Linear branch are, so the state_dict keys are
        # time_mlp.{1,3} / cond_mlp.{1,3} instead of time_mlp.{1,3,5} / cond_mlp.{0,2,4}
        if time_embed:
            time_layers = [_Holder()]
        else:
            time_layers = [nn.Unflatten(-1, (-1, 1)), nn.Linear(1, half // 2), nn.GELU()]
        self.time_mlp = nn.Sequential(*time_layers, nn.Linear(half // 2, half), nn.GELU(), nn.Linear(half, half))
        hidden = max(cond_size, half // 2)
        cond_layers = [_Holder()] if cond_embed else [nn.Linear(cond_size, hidden), nn.GELU()]
        self.cond_mlp = nn.Sequential(*cond_layers, nn.Linear(hidden, half), nn.GELU(), nn.Linear(half, half))

        self.downs, self.ups = nn.ModuleList([]), nn.ModuleList([])
        self.downs_attn, self.ups_attn = nn.ModuleList([]), nn.ModuleList([])
        pairs = list(zip(layer_sizes[:-1], layer_sizes[1:]))
        nres = len(pairs)

        shape = self.grid
        self.level_shapes = [shape]
        extras: List[List[int]] = []
        for lv, (cin, cout) in enumerate(pairs):
            last = lv == nres - 1
            if not last:
                extras.append([(shape[0] + 1) % 2, shape[1] % 2, shape[2] % 2])
                shape = (math.ceil(shape[0] / 2.0) if compress_Z else shape[0], shape[1] // 2, shape[2] // 2)
                self.level_shapes.append(shape)
            self.downs.append(nn.ModuleList([
                _Resnet(cin, cout, cond_dim, g),
                _Resnet(cout, cout, cond_dim, g),
                _CylConv(cout, cout, (3, 4, 4), stride=(zs, 2, 2), pad_zr=1) if not last else nn.Identity(),
            ]))
            if block_attn:
                self.downs_attn.append(_attn(cout))

        mid = layer_sizes[-1]
        self.mid_block1 = _Resnet(mid, mid, cond_dim, g)
        self.mid_attn = _attn(mid) if mid_attn else False
        self.mid_block2 = _Resnet(mid, mid, cond_dim, g)

        # per up level: z kernel extent and (z, phi, r) output padding of the transposed conv
        self.up_kernel_z: List[int] = []
        self.up_out_pad: List[Sequence[int]] = []
        for lv, (cin, cout) in enumerate(reversed(pairs)):
            last = lv == nres - 1
            up = nn.Identity()
            r1 = _Resnet(cout * 2, cin, cond_dim, g)
            r2 = _Resnet(cin, cin, cond_dim, g)
            if not last:
                e = extras.pop()
                kz = 4 if e[0] > 0 else 3
                opad = (0, e[1], e[2])
                self.up_kernel_z.append(kz)
                self.up_out_pad.append(opad)
                up = _CylConvT(cin, (kz, 4, 4), (zs, 2, 2), opad)
            self.ups.append(nn.ModuleList([r1, r2, up]))
            if block_attn:
                self.ups_attn.append(_attn(cin))

        head = _CylConv(layer_sizes[0], out_dim, 1)
        self.final_conv = nn.Sequential(_Resnet(layer_sizes[1], layer_sizes[0], None, g), head)
        self._engine = None

    # ------------------------------------------------------------------ HIP path
    def engine(self):
        """The HIP plan bound to this parameter set (created on first use; needs a GPU)."""
        if self._engine is None:
            from .engine import UnetEngine
            self._engine = UnetEngine.for_unet(self)
        return self._engine

    def forward(self, x, cond=None, time=None, controls=None):
        """Same contract as reference CondUnet.forward (models.py:701-748): x (B,C,D,H,W) fp32 -> (B,1,D,H,W)."""
        if controls is not None:
            raise NotImplementedError("ControlNet hidden-state injection is dead code in the reference (SURVEY 2 #8)")
        return self.engine().unet_forward(x, cond, time)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = None  # parameters may have moved: rebind on next use
        return out


def unet_kwargs_from_config(cfg: dict) -> dict:
    """Argument derivation of reference CaloDiffusion.init_model (models/calodiffusion.py:39-81)."""
    channels = 3 if cfg.get("R_Z_INPUT", False) else 1
    if cfg.get("PHI_INPUT", False):
        channels += 1
    cond_size = 2 + cfg["SHAPE_FINAL"][2] if "layer" in cfg.get("SHOWERMAP", "") else 1
    if cfg.get("HGCAL", False):
        cond_size += 2
    return dict(
        cond_dim=cfg["COND_SIZE_UNET"],
        out_dim=1,
        channels=channels,
        layer_sizes=list(cfg["LAYER_SIZE_UNET"]),
        block_attn=cfg.get("BLOCK_ATTN", False),
        mid_attn=cfg.get("MID_ATTN", False),
        cylindrical=cfg.get("CYLINDRICAL", False),
        compress_Z=cfg.get("COMPRESS_Z", False),
        resnet_block_groups=cfg.get("BLOCK_GROUPS", 8),
        data_shape=[1, channels] + list(cfg["SHAPE_FINAL"][1:]),
        cond_embed=(cfg.get("COND_EMBED", "sin") == "sin"),
        cond_size=cond_size,
        time_embed=(cfg.get("TIME_EMBED", "sin") == "sin"),
    )
