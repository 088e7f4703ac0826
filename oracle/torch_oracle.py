"""CPU oracle for the CaloDiffusion denoising hot path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement of the reference
algorithm (OzAmram/CaloDiffusion, mounted read-only at /root/reference when the
fixtures were generated).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker /
the timed CPU baseline -- never as a product path.  The product
(``calodiffusion_amd``) fails loudly when its HIP library is missing; it never
falls back to this module.

Arithmetic boundary: like the reference, the dense arithmetic (conv3d,
conv_transpose3d, group_norm, linear, softmax, einsum) is done by stock PyTorch
CPU kernels (the reference un-pins ``torch>=2.0``, pyproject.toml:14).  What is
restated here is everything the reference builds on top of them: the cylindrical
padding rule, the U-Net wiring, the EDM pre-conditioning, the cosine schedule,
the DDIM/DDPM update and the hybrid-weight loss.  The restatement is functional
(it walks a flat ``state_dict``) and shares no code with the reference.

Parity pin: ``tests/golden/*.npz`` were produced by importing the *reference's
own* classes in the build container (``oracle/gen_golden.py``); ``tests/
test_oracle_golden.py`` checks this file against them.

Layout: NCDHW fp32, D = z (layer), H = phi (periodic), W = r
(reference calodiffusion/models/models.py:26,66).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# schedules  (reference calodiffusion/utils/sampling.py:16-24, models/sample.py:45-54)
# --------------------------------------------------------------------------
def cosine_beta_schedule(nsteps: int, s: float = 0.008) -> Tensor:
    """fp32 cosine schedule, same op order as utils/sampling.py:16-24."""
    grid = torch.linspace(0, nsteps, nsteps + 1)
    abar = torch.cos(((grid / nsteps) + s) / (1 + s) * np.pi * 0.5) ** 2
    abar = abar / abar[0]
    betas = 1 - (abar[1:] / abar[:-1])
    return torch.clip(betas, 0.0001, 0.9999)


@dataclass
class StepTables:
    """Per-step scalars of the DDIM/DDPM loop (models/sample.py:45-101)."""

    alphas_cumprod: Tensor
    alphas_cumprod_prev: Tensor
    sqrt_alphas_cumprod: Tensor
    sqrt_one_minus_alphas_cumprod: Tensor


def ddim_tables(num_steps: int) -> StepTables:
    betas = cosine_beta_schedule(num_steps)
    ac = torch.cumprod(1.0 - betas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    return StepTables(ac, ac_prev, torch.sqrt(ac), torch.sqrt(1.0 - ac))


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def cyl_conv3d(x: Tensor, w: Tensor, b: Optional[Tensor], stride=(1, 1, 1), padding=(0, 0, 0)) -> Tensor:
    """phi-periodic Conv3d (models/models.py:65-96).

    ``padding`` is the *requested* (z, phi, r) padding; phi is wrapped circularly
    by ``padding[1]`` on each side, z and r are zero padded.
    """
    pz, pp, pr = padding
    if pp > 0:
        x = F.pad(x, (0, 0, pp, pp, 0, 0), mode="circular")
    return F.conv3d(x, w, b, stride=stride, padding=(pz, 0, pr))


def cyl_conv_transpose3d(x: Tensor, w: Tensor, b: Optional[Tensor], stride, output_padding) -> Tensor:
    """phi-periodic ConvTranspose3d used by Upsample (models/models.py:25-62, 335-348).

    Requested padding is 1 everywhere; the phi halo of 1 added by the circular pad
    is cancelled by a transposed-conv padding of kH-1 (models/models.py:45).
    """
    kh = w.shape[3]
    x = F.pad(x, (0, 0, 1, 1, 0, 0), mode="circular")
    return F.conv_transpose3d(x, w, b, stride=stride, padding=(1, kh - 1, 1), output_padding=tuple(output_padding))


def _conv(sd: SD, name: str, x: Tensor, cyl: bool, stride=(1, 1, 1), padding=(0, 0, 0)) -> Tensor:
    """Conv wrapper: cylindrical modules keep their Conv3d under ``.conv``."""
    if cyl:
        return cyl_conv3d(x, sd[name + ".conv.weight"], sd.get(name + ".conv.bias"), stride, padding)
    return F.conv3d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def block(sd: SD, p: str, x: Tensor, groups: int, cyl: bool) -> Tensor:
    """conv3x3x3 -> GroupNorm -> SiLU  (models/models.py:147-169; scale_shift never passed)."""
    x = _conv(sd, p + ".proj", x, cyl, padding=(1, 1, 1))
    x = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
    return F.silu(x)


def resnet_block(sd: SD, p: str, x: Tensor, cond: Optional[Tensor], groups: int, cyl: bool) -> Tensor:
    """models/models.py:172-200."""
    h = block(sd, p + ".block1", x, groups, cyl)
    if cond is not None and (p + ".mlp.1.weight") in sd:
        t = F.linear(F.silu(cond), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])
        h = h + t[:, :, None, None, None]
    h = block(sd, p + ".block2", h, groups, cyl)
    if (p + ".res_conv.conv.weight") in sd or (p + ".res_conv.weight") in sd:
        return h + _conv(sd, p + ".res_conv", x, cyl)
    return h + x


def linear_attention(sd: SD, p: str, x: Tensor, cyl: bool, heads: int = 1, dim_head: int = 32) -> Tensor:
    """models/models.py:281-318 (heads=1, dim_head=32 are never overridden)."""
    b, c, d, h, w = x.shape
    qkv = _conv(sd, p + ".to_qkv", x, cyl)
    q, k, v = (t.reshape(b, heads, dim_head, d * h * w) for t in qkv.chunk(3, dim=1))
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q).reshape(b, heads * dim_head, d, h, w)
    out = _conv(sd, p + ".to_out.0", out, cyl)
    return F.group_norm(out, 1, sd[p + ".to_out.1.weight"], sd[p + ".to_out.1.bias"], eps=1e-5)


def attn_residual(sd: SD, p: str, x: Tensor, cyl: bool) -> Tensor:
    """Residual(PreNorm(LinearAttention))  (models/models.py:111-117, 321-329)."""
    xn = F.group_norm(x, 1, sd[p + ".fn.norm.weight"], sd[p + ".fn.norm.bias"], eps=1e-5)
    return linear_attention(sd, p + ".fn.fn", xn, cyl) + x


def _mlp(sd: SD, p: str, x: Tensor, idx: Sequence[int]) -> Tensor:
    """Linear/GELU stack with exact-erf GELU; ``idx`` are the nn.Sequential slots of the Linears."""
    for n, i in enumerate(idx):
        x = F.linear(x, sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"])
        if n + 1 < len(idx):
            x = F.gelu(x)
    return x


# --------------------------------------------------------------------------
# U-Net description (mirrors CondUnet.__init__, models/models.py:525-699)
# --------------------------------------------------------------------------
@dataclass
class UnetSpec:
    layer_sizes: List[int]
    channels: int
    cond_dim: int = 128
    cond_size: int = 1
    groups: int = 8
    block_attn: bool = True
    mid_attn: bool = True
    compress_z: bool = True
    cylindrical: bool = True
    data_shape: Tuple[int, int, int] = (45, 16, 9)
    time_sin: bool = False  # SinusoidalPositionEmbeddings instead of the first Linear of the time / cond MLP
    cond_sin: bool = False  # (models/models.py:578-601; reachable through CondUnet.forward only)
    # derived
    level_shapes: List[Tuple[int, int, int]] = field(default_factory=list)
    up_kernel_z: List[int] = field(default_factory=list)
    up_out_pad: List[Tuple[int, int, int]] = field(default_factory=list)

    def __post_init__(self):
        shape = tuple(self.data_shape)
        self.level_shapes = [shape]
        extras = []
        nres = len(self.layer_sizes) - 1
        for ind in range(nres - 1):
            extras.append(((shape[0] + 1) % 2, shape[1] % 2, shape[2] % 2))
            zd = math.ceil(shape[0] / 2.0) if self.compress_z else shape[0]
            shape = (zd, shape[1] // 2, shape[2] // 2)
            self.level_shapes.append(shape)
        # ups consume the extras in reverse; Upsample zeroes element 0 after
        # choosing the z-kernel (models/models.py:335-339)
        self.up_kernel_z, self.up_out_pad = [], []
        for e in reversed(extras):
            self.up_kernel_z.append(4 if e[0] > 0 else 3)
            self.up_out_pad.append((0, e[1], e[2]))

    @property
    def z_stride(self) -> int:
        return 2 if self.compress_z else 1


def spec_from_config(cfg: dict) -> UnetSpec:
    """Argument derivation of CaloDiffusion.init_model (models/calodiffusion.py:39-81)."""
    ch = 1
    if cfg.get("R_Z_INPUT", False):
        ch = 3
    if cfg.get("PHI_INPUT", False):
        ch += 1
    cond_size = 2 + cfg["SHAPE_FINAL"][2] if "layer" in cfg.get("SHOWERMAP", "") else 1
    if cfg.get("HGCAL", False):
        cond_size += 2
    if cfg.get("COND_EMBED", "sin") == "sin" or cfg.get("TIME_EMBED", "sin") == "sin":
        raise NotImplementedError("sinusoidal embeddings are dormant in every shipped config")
    return UnetSpec(
        layer_sizes=list(cfg["LAYER_SIZE_UNET"]),
        channels=ch,
        cond_dim=cfg["COND_SIZE_UNET"],
        cond_size=cond_size,
        groups=cfg.get("BLOCK_GROUPS", 8),
        block_attn=cfg.get("BLOCK_ATTN", False),
        mid_attn=cfg.get("MID_ATTN", False),
        compress_z=cfg.get("COMPRESS_Z", False),
        cylindrical=cfg.get("CYLINDRICAL", False),
        data_shape=tuple(cfg["SHAPE_FINAL"][2:]),
    )


def sinusoidal_embedding(v: Tensor, dim: int) -> Tensor:
    """SinusoidalPositionEmbeddings.forward (models/models.py:132-144) of a (B,) tensor."""
    half = dim // 2
    freq = torch.exp(torch.arange(half) * -(np.log(10000) / (half - 1)))
    arg = v[:, None] * freq[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def cond_unet_forward(sd: SD, spec: UnetSpec, x: Tensor, cond: Tensor, time: Tensor) -> Tensor:
    """CondUnet.forward (models/models.py:701-748); embeddings: Linear branch (:583-587, :601) or sinusoidal (:581, :599)."""
    cyl, g = spec.cylindrical, spec.groups
    x = _conv(sd, "init_conv", x, cyl, padding=(1, 1, 1))
    q = spec.cond_dim // 4
    c = _mlp(sd, "cond_mlp", sinusoidal_embedding(cond.reshape(-1), q), (1, 3)) if spec.cond_sin else _mlp(sd, "cond_mlp", cond, (0, 2, 4))
    t = _mlp(sd, "time_mlp", sinusoidal_embedding(time.reshape(-1), q), (1, 3)) if spec.time_sin else \
        _mlp(sd, "time_mlp", time.reshape(-1, 1), (1, 3, 5))
    conditions = torch.cat([t, c], dim=-1)

    nres = len(spec.layer_sizes) - 1
    zs = spec.z_stride
    skips = []
    for i in range(nres):
        x = resnet_block(sd, f"downs.{i}.0", x, conditions, g, cyl)
        x = resnet_block(sd, f"downs.{i}.1", x, conditions, g, cyl)
        if spec.block_attn:
            x = attn_residual(sd, f"downs_attn.{i}", x, cyl)
        skips.append(x)
        if i < nres - 1:
            x = _conv(sd, f"downs.{i}.2", x, cyl, stride=(zs, 2, 2), padding=(1, 1, 1))

    x = resnet_block(sd, "mid_block1", x, conditions, g, cyl)
    if spec.mid_attn:
        x = attn_residual(sd, "mid_attn", x, cyl)
    x = resnet_block(sd, "mid_block2", x, conditions, g, cyl)

    for i in range(nres):
        x = torch.cat((x, skips.pop()), dim=1)
        x = resnet_block(sd, f"ups.{i}.0", x, conditions, g, cyl)
        x = resnet_block(sd, f"ups.{i}.1", x, conditions, g, cyl)
        if spec.block_attn:
            x = attn_residual(sd, f"ups_attn.{i}", x, cyl)
        if i < nres - 1:
            wt, bs = sd[f"ups.{i}.2.convTrans.weight"], sd[f"ups.{i}.2.convTrans.bias"]
            if cyl:
                x = cyl_conv_transpose3d(x, wt, bs, (zs, 2, 2), spec.up_out_pad[i])
            else:
                x = F.conv_transpose3d(x, wt, bs, stride=(zs, 2, 2), padding=1, output_padding=spec.up_out_pad[i])

    x = resnet_block(sd, "final_conv.0", x, None, g, cyl)
    return _conv(sd, "final_conv.1", x, cyl)


# --------------------------------------------------------------------------
# coordinate images (utils/utils.py:33-150)
# --------------------------------------------------------------------------
_R_EDGES = {
    2: [0, 4.65, 9.3, 13.95, 18.6, 23.25, 27.9, 32.55, 37.2, 41.85],
    3: [0, 2.325, 4.65, 6.975, 9.3, 11.625, 13.95, 16.275, 18.6, 20.925, 23.25, 25.575, 27.9, 30.225, 32.55,
        34.875, 37.2, 39.525, 41.85],
    1: [0.0, 2.0, 4.0, 5.0, 6.0, 8.0, 10.0, 12.0, 15.0, 20.0, 25.0, 30.0, 40.0, 50.0, 60.0, 70.0, 80.0, 90.0,
        100.0, 120.0, 130.0, 150.0, 160.0, 200.0, 250.0, 300.0, 350.0, 400.0, 600.0, 1000.0, 2000.0],
    0: [0.00, 1.00, 4.00, 5.00, 7.00, 10.00, 15.00, 20.00, 30.00, 50.00, 80.00, 90.00, 100.00, 130.00, 150.00,
        160.00, 200.00, 250.00, 300.00, 350.00, 400.00, 600.00, 1000.00, 2000.00],
}


def r_bin_edges(dataset_num: int, n_r: int) -> List[float]:
    if dataset_num >= 100:  # HGCal: unit-width bins (utils/utils.py:126-127)
        return [float(i) for i in range(n_r + 1)]
    return list(_R_EDGES[dataset_num])


def rz_phi_profiles(dataset_num: int, shape_dhw: Sequence[int]) -> Tuple[Tensor, Tensor, Tensor]:
    """1-D profiles of the constant R (over r), Z (over z) and phi (over phi) input channels.

    create_R_Z_image(scaled=True) normalises R by the last bin centre and Z by the
    layer count (utils/utils.py:131-149); phi is linspace(0,1,n_phi) (:33-39).
    """
    nz, nphi, nr = shape_dhw
    edges = r_bin_edges(dataset_num, nr)
    centres = [(edges[i] + edges[i + 1]) / 2.0 for i in range(len(edges) - 1)]
    if len(centres) != nr:
        raise ValueError(f"Mismatch for dataset size {tuple(shape_dhw)} and dataset num {dataset_num}")
    # the reference writes python floats into an fp32 image, then divides in fp32
    r = torch.tensor(centres, dtype=torch.float32) / centres[-1]
    z = torch.arange(nz, dtype=torch.float32) / nz
    phi = torch.linspace(0.0, 1.0, nphi, dtype=torch.float32)
    return r, z, phi


def add_rz_phi(x: Tensor, dataset_num: int, r_z: bool, phi_in: bool) -> Tensor:
    """CaloDiffusion.add_RZPhi (models/calodiffusion.py:121-142)."""
    b, _, nz, nphi, nr = x.shape
    r, z, phi = rz_phi_profiles(dataset_num, (nz, nphi, nr))
    cats = [x]
    if r_z:
        cats.append(r.view(1, 1, 1, 1, nr).expand(b, 1, nz, nphi, nr))
        cats.append(z.view(1, 1, nz, 1, 1).expand(b, 1, nz, nphi, nr))
    if phi_in:
        cats.append(phi.view(1, 1, 1, nphi, 1).expand(b, 1, nz, nphi, nr))
    return torch.cat(cats, dim=1) if len(cats) > 1 else x


# --------------------------------------------------------------------------
# diffusion wrapper
# --------------------------------------------------------------------------
def sigma_data_of(cfg: dict) -> float:
    """models/loss.py:14-25."""
    return 1.0 if "log" in cfg.get("NOISE_SCHED", "linear") else 0.5


def edm_scalings(sigma: Tensor, sigma_data: float):
    """Loss.get_scaling (models/loss.py:29-41) -> (c_skip, c_out, c_in)."""
    s2 = sigma ** 2 + sigma_data ** 2
    return sigma_data ** 2 / s2, sigma * sigma_data / s2 ** 0.5, 1 / s2 ** 0.5


def time_embed(sigma: Tensor, kind: str) -> Tensor:
    """CaloDiffusion.do_time_embed (models/calodiffusion.py:144-152)."""
    if kind == "sigma":
        return sigma / (1 + sigma ** 2).sqrt()
    if kind == "log":
        return 0.5 * torch.log(sigma)
    raise KeyError(kind)


class OracleModel:
    """Functional stand-in for reference CaloDiffusion (models/calodiffusion.py:9-173).

    ``sd`` holds CondUnet weights keyed as in the reference, with or without the
    ``model.`` prefix.
    """

    def __init__(self, cfg: dict, sd: SD):
        self.cfg = cfg
        self.spec = spec_from_config(cfg)
        self.sd = {(k[6:] if k.startswith("model.") else k): v for k, v in sd.items()}
        self.sigma_data = sigma_data_of(cfg)
        self.time_kind = cfg.get("TIME_EMBED", "sin")
        self.layer_cond = "layer" in cfg.get("SHOWERMAP", "")
        self.objective = cfg.get("TRAINING_OBJ", "noise_pred")
        self.dataset_num = cfg.get("DATASET_NUM", 2)

    def forward(self, x: Tensor, E: Tensor, t_emb: Tensor, layers: Optional[Tensor]) -> Tensor:
        """models/calodiffusion.py:86-98 (NN_embed inactive on the regular grid)."""
        cond = torch.cat([E, layers], dim=1) if (self.layer_cond and layers is not None) else E
        xin = add_rz_phi(x, self.dataset_num, self.cfg.get("R_Z_INPUT", False), self.cfg.get("PHI_INPUT", False))
        return cond_unet_forward(self.sd, self.spec, xin.float(), cond.float(), t_emb.float())

    def denoise(self, x: Tensor, E: Tensor, sigma: Tensor, layers: Optional[Tensor]) -> Tensor:
        """models/calodiffusion.py:154-169."""
        sigma = sigma.reshape(-1, 1, 1, 1, 1)
        t_emb = time_embed(sigma.reshape(-1), self.time_kind)
        c_skip, c_out, c_in = edm_scalings(sigma, self.sigma_data)
        pred = self.forward(x * c_in, E, t_emb, layers)
        if "noise_pred" in self.objective:
            return x - sigma * pred
        if "mean_pred" in self.objective:
            return pred
        if "hybrid" in self.objective:
            return c_skip * x + c_out * pred
        raise ValueError("??? Training obj %s" % self.objective)

    @torch.no_grad()
    def ddim_sample(self, start: Tensor, E: Tensor, layers: Optional[Tensor], num_steps: int, eta: float = 0.0,
                    sample_offset: int = 0, step_noise: Optional[Sequence[Tensor]] = None,
                    keep: bool = False, stop_after: Optional[int] = None):
        """DDim.__call__ / DDPM (models/sample.py:41-121).

        ``step_noise[i]`` is the noise tensor of loop iteration i (only read when eta != 0;
        the reference draws one per step regardless).
        """
        tb = ddim_tables(num_steps)
        steps = list(range(num_steps - 1, -1, -1))[sample_offset:]
        t0 = steps[0]
        x = start * (tb.sqrt_one_minus_alphas_cumprod[t0] / tb.sqrt_alphas_cumprod[t0])
        xs, x0s = [], []
        B = start.shape[0]
        for i, t in enumerate(steps):
            if stop_after is not None and i >= stop_after:  # (tests: the first part of a long trajectory)
                break
            a, a_prev = tb.alphas_cumprod[t], tb.alphas_cumprod_prev[t]
            denom = tb.sqrt_alphas_cumprod[max(t - 1, 0)]
            sigma = tb.sqrt_one_minus_alphas_cumprod[t] / tb.sqrt_alphas_cumprod[t]
            x0 = self.denoise(x, E, sigma.expand(B), layers)
            eps = (x - x0) / sigma
            ddim_sigma = eta * (((1 - a_prev) / (1 - a)) * (1 - a / a_prev)) ** 0.5
            sigma_prev = (1.0 - a_prev - ddim_sigma ** 2).sqrt() / denom
            x = x0 + (sigma_prev * eps if t > 0 else 0.0)
            if eta != 0.0:
                x = x + ddim_sigma * step_noise[i] / denom
            if keep:
                xs.append(x)
                x0s.append(x0)
        return x, xs, x0s

    def edm_euler_sample(self, start: Tensor, E: Tensor, layers: Optional[Tensor], num_steps: int, sample_offset: int = 0,
                         sigma_min: float = 0.002, sigma_max: float = 80.0, rho: float = 7, keep: bool = False):
        """EDMAbstract.setup / for_loop + Euler.in_loop_sampler with S_churn = 0 (models/sample.py:639-727, 771-789)."""
        step_indices = torch.arange(num_steps, dtype=torch.float32)
        t_steps = (sigma_max ** (1 / rho) + step_indices / (num_steps - 1)
                   * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
        t_steps = torch.cat([t_steps, torch.zeros_like(t_steps[:1])])[sample_offset:]
        x_next = start.to(torch.float32) * t_steps[0]
        xs, x0s = [], []
        B = start.shape[0]
        for t_cur, t_next in zip(t_steps[:-1], t_steps[1:]):
            x_hat = x_next  # gamma = 0: t_hat = t_cur and the churn noise is multiplied by sqrt(0)
            denoised = self.denoise(x_hat, E, t_cur.expand(B), layers)
            d_cur = (x_hat - denoised) / t_cur
            x_next = x_hat + (t_next - t_cur) * d_cur
            if keep:
                xs.append(x_hat)
                x0s.append(denoised)
        return x_next, xs, x0s

    def hybrid_l2_loss(self, data: Tensor, E: Tensor, noise: Tensor, layers: Optional[Tensor],
                       rnd_normal: Optional[Tensor] = None, time: Optional[Tensor] = None,
                       n_steps: int = 400, loss_type: str = "l2") -> Tensor:
        """Loss.__call__ + the objective class named by TRAINING_OBJ (hybrid_weight / noise_pred / mean_pred) + the reduction of
        Loss._loss (models/loss.py:97-116, 118-142, 163-210): 'l2' (weighted), 'l1', 'mse', 'huber' (= smooth_l1_loss, beta 1)."""
        shp = (data.shape[0], 1, 1, 1, 1)
        if "log" in self.cfg.get("NOISE_SCHED", "linear"):
            sigma = (rnd_normal * 1.2 + (-1.2)).exp().reshape(shp)
        else:
            tb = ddim_tables(n_steps)
            sigma = (tb.sqrt_one_minus_alphas_cumprod[time] / tb.sqrt_alphas_cumprod[time]).reshape(shp)
        x0 = self.denoise(data + sigma * noise, E, sigma, layers)
        # the three objective classes (self.objective = the loss class name, as the reference's denoise branches on it)
        if "noise_pred" in self.objective:      # models/loss.py:181-196
            x0_pred = data - sigma * x0
            pred, target, w = (data - x0_pred) / sigma, noise, torch.ones_like(x0)
        elif "mean_pred" in self.objective:     # models/loss.py:198-210
            pred, target, w = x0, data, 1.0 / (sigma ** 2)
        else:                                   # hybrid_weight, models/loss.py:163-179
            pred, target, w = x0, data, (1.0 + 1.0 / sigma ** 2).reshape(shp)
        if loss_type == "l1":
            return F.l1_loss(pred, target)
        if loss_type == "mse":
            return F.mse_loss(pred, target)
        if loss_type == "huber":
            return F.smooth_l1_loss(pred, target)
        return (w * (pred - target) ** 2).sum() / (torch.mean(w) * float(np.prod(data.shape)))


# --------------------------------------------------------------------------
# LayerDiffusion's layer-energy model (models/models.py:373-457, models/layerdiffusion.py:109-132)
# --------------------------------------------------------------------------
def resnet_mlp_forward(sd: SD, x: Tensor, cond: Tensor, time: Tensor) -> Tensor:
    """ResNet.forward (models/models.py:444-457) with ResDense.forward (models.py:385-391)."""
    c = _mlp(sd, "cond_mlp", cond, (0, 2, 4))
    t = _mlp(sd, "time_mlp", time.reshape(-1, 1), (1, 3, 5))
    emb = torch.cat([c, t], dim=-1)
    x = F.linear(x, sd["in_lay.weight"], sd["in_lay.bias"])
    i = 0
    while f"hidden_layers.{i}.dense1.0.weight" in sd:
        p = f"hidden_layers.{i}"
        h = F.gelu(F.linear(x, sd[p + ".dense1.0.weight"], sd[p + ".dense1.0.bias"]))
        h = h + F.linear(F.gelu(emb), sd[p + ".embeder.1.weight"], sd[p + ".embeder.1.bias"])
        h = F.gelu(F.linear(h, sd[p + ".dense2.0.weight"], sd[p + ".dense2.0.bias"]))
        x = h + x
        i += 1
    return F.linear(x, sd["out_lay.weight"], sd["out_lay.bias"])


class OracleLayerModel(OracleModel):
    """The layer stage of reference LayerDiffusion (set_layer_state(True)): the denoiser and the sampler loops of OracleModel
    on (B, D+1) vectors with the ResNet MLP as the network; `layers` is ignored (layerdiffusion.py:109-112)."""

    def __init__(self, cfg: dict, sd: SD):
        self.cfg = cfg
        self.sd = {(k[12:] if k.startswith("layer_model.") else k): v for k, v in sd.items()}
        self.sigma_data = sigma_data_of(cfg)
        self.time_kind = cfg.get("TIME_EMBED", "sin")
        self.objective = cfg.get("TRAINING_OBJ", "noise_pred")

    def forward(self, x: Tensor, E: Tensor, t_emb: Tensor, layers: Optional[Tensor] = None) -> Tensor:
        return resnet_mlp_forward(self.sd, x.float(), E.float(), t_emb.float())

    def denoise(self, x: Tensor, E: Tensor, sigma: Tensor, layers: Optional[Tensor] = None) -> Tensor:
        sigma = sigma.reshape(-1, 1)
        t_emb = time_embed(sigma.reshape(-1), self.time_kind)
        c_skip, c_out, c_in = edm_scalings(sigma, self.sigma_data)
        pred = self.forward(x * c_in, E, t_emb)
        if "noise_pred" in self.objective:
            return x - sigma * pred
        if "mean_pred" in self.objective:
            return pred
        return c_skip * x + c_out * pred

    def hybrid_l2_loss(self, data: Tensor, E: Tensor, noise: Tensor, layers: Optional[Tensor] = None,
                       rnd_normal: Optional[Tensor] = None, time: Optional[Tensor] = None, n_steps: int = 400,
                       loss_type: str = "l2") -> Tensor:
        """Loss.__call__ + hybrid_weight + the reduction of Loss._loss on (B, D+1) layer vectors (models/loss.py:97-116,118-142,
        163-179), as LayerDiffusion.compute_loss applies it in the layer state (layerdiffusion.py:52-57)."""
        B = data.shape[0]
        if "log" in self.cfg.get("NOISE_SCHED", "linear"):
            sigma = (rnd_normal * 1.2 + (-1.2)).exp().reshape(B, 1)
        else:
            tb = ddim_tables(n_steps)
            sigma = (tb.sqrt_one_minus_alphas_cumprod[time] / tb.sqrt_alphas_cumprod[time]).reshape(B, 1)
        x0 = self.denoise(data + sigma * noise, E, sigma)
        if loss_type == "l1":
            return F.l1_loss(x0, data)
        if loss_type == "mse":
            return F.mse_loss(x0, data)
        if loss_type == "huber":
            return F.smooth_l1_loss(x0, data)
        w = 1.0 + 1.0 / sigma ** 2
        return (w * (x0 - data) ** 2).sum() / (torch.mean(w) * float(np.prod(data.shape)))


# --------------------------------------------------------------------------
# work accounting: algorithmic FLOPs / bytes per sample-step
# --------------------------------------------------------------------------
def algorithmic_work(spec: UnetSpec) -> Dict[str, float]:
    """FLOPs (2/MAC) and fused-fp32 bytes per sample per denoise step, split by op family."""
    L = spec.layer_sizes
    nres = len(L) - 1
    vox = [s[0] * s[1] * s[2] for s in spec.level_shapes]
    fl = {"conv3": 0.0, "conv1": 0.0, "updown": 0.0, "attn": 0.0, "linear": 0.0}
    by = {"conv3": 0.0, "conv1": 0.0, "updown": 0.0}

    def c3(ci, co, n):
        fl["conv3"] += 2.0 * 27 * ci * co * n
        by["conv3"] += 4.0 * n * (ci + co)

    def c1(ci, co, n):
        fl["conv1"] += 2.0 * ci * co * n
        by["conv1"] += 4.0 * n * (ci + co)

    def res(ci, co, n):
        c3(ci, co, n)
        c3(co, co, n)
        if ci != co:
            c1(ci, co, n)

    def attn(c, n):
        c1(c, 96, n)
        c1(32, c, n)
        fl["attn"] += 2.0 * 2 * 32 * 32 * n

    c3(spec.channels, L[0], vox[0])
    for i in range(nres):
        res(L[i], L[i + 1], vox[i]); res(L[i + 1], L[i + 1], vox[i])
        if spec.block_attn:
            attn(L[i + 1], vox[i])
        if i < nres - 1:
            fl["updown"] += 2.0 * 48 * L[i + 1] * L[i + 1] * vox[i + 1]
            by["updown"] += 4.0 * L[i + 1] * (vox[i] + vox[i + 1])
    res(L[-1], L[-1], vox[-1]); res(L[-1], L[-1], vox[-1])
    if spec.mid_attn:
        attn(L[-1], vox[-1])
    for i, (ci, co) in enumerate(reversed(list(zip(L[:-1], L[1:])))):
        lv = nres - 1 - i
        res(co * 2, ci, vox[lv]); res(ci, ci, vox[lv])
        if spec.block_attn:
            attn(ci, vox[lv])
        if i < nres - 1:
            kz = spec.up_kernel_z[i]
            fl["updown"] += 2.0 * kz * 16 * ci * ci * vox[lv]
            by["updown"] += 4.0 * ci * (vox[lv] + vox[lv - 1])
    res(L[1], L[0], vox[0])
    c1(L[0], 1, vox[0])
    fl["total"] = sum(fl.values())
    by["total"] = sum(by.values()) + 3 * 4.0 * vox[0]
    return {"flops": fl, "bytes": by}


# ------------------------------------------------------------------------------------------------------------------
# Inverse pre-processing (utils.ReverseNormCaloChall, calodiffusion/utils/utils.py:446-573) for the regular grids
# ------------------------------------------------------------------------------------------------------------------
def reverse_norm_calochall(voxels, e, layerE, consts, emax=1000.0, emin=1.0, max_deposit=2, logE=True, ecut=0.0):
    """numpy restatement; ``consts`` = the dataset's normalisation constants (utils/consts.py:82-116);
    layer mode ('layer-logit-norm') when layerE is given, 'logit-norm' otherwise.  Returns (data (B, -1), energy)."""
    import numpy as np

    def reverse_logit(x, alpha=1e-6):  # utils.py:233-237
        ex = np.exp(x)
        o = ex / (1 + ex)
        return (o - alpha) / (1 - 2 * alpha)

    energy = emin * (emax / emin) ** e if logE else emin + (emax - emin) * e
    voxels = (voxels * consts["logit_std"]) + consts["logit_mean"]
    data = reverse_logit(voxels)
    if layerE is not None:
        totalE, layers = layerE[:, :1], layerE[:, 1:]
        totalE = (totalE * consts["totalE_std"]) + consts["totalE_mean"]
        layers = reverse_logit((layers * consts["layers_std"]) + consts["layers_mean"])
        layers = layers / np.sum(layers, axis=1, keepdims=True) * totalE
        data = np.squeeze(data, axis=1).copy()
        data[data < 0] = 0
        prev = np.sum(data, (2, 3), keepdims=True)
        layers = layers.reshape((-1, data.shape[1], 1, 1))
        fac = layers / (prev + 1e-10)
        fac[layers < 1e-6] = 1.0
        fac[prev < 1e-6] = 1.0
        data = data * fac
    data = data.reshape(voxels.shape[0], -1) * max_deposit * energy.reshape(-1, 1)
    if ecut > 0:
        data[data < ecut] = 0
    return data, energy


def reverse_norm_hgcal(voxels, e, layerE, consts, emax=1000.0, emin=1.0, max_deposit=2, decode=None):
    """utils.ReverseNormHGCal (utils/HGCal_utils.py:167-292) for the [layer-]logit-norm maps, numpy: energy linear in e,
    reverse_logit with alpha 1e-8, `decode` (optional) = the geometry decode between the inverse logit and the layer
    renormalisation, "essentially zero" = 1e-8, the energy cut disabled."""
    def reverse_logit(x, alpha=1e-8):  # HGCal_utils.py:13-17
        ex = np.exp(x)
        o = ex / (1 + ex)
        return (o - alpha) / (1 - 2 * alpha)

    gen_out = np.array(emin) + (np.array(emax) - np.array(emin)) * e
    energy = gen_out[:, 0]
    data = reverse_logit((voxels * consts["logit_std"]) + consts["logit_mean"])
    if decode is not None:
        data = decode(data)
    if layerE is not None:
        totalE, layers = layerE[:, :1], layerE[:, 1:]
        totalE = (totalE * consts["totalE_std"]) + consts["totalE_mean"]
        layers = reverse_logit((layers * consts["layers_std"]) + consts["layers_mean"])
        layers = layers / np.sum(layers, axis=1, keepdims=True) * totalE
        data = np.squeeze(data).copy()
        data[data < 0] = 0
        prev = np.sum(data, (2), keepdims=True)
        layers = layers.reshape((-1, data.shape[1], 1))
        fac = layers / (prev + 1e-10)
        fac[layers < 1e-8] = 1.0
        fac[prev < 1e-8] = 1.0
        data = data * fac
        return data * max_deposit * energy.reshape(-1, 1, 1), gen_out
    return data * max_deposit * energy.reshape((-1,) + (1,) * (data.ndim - 1)), gen_out
