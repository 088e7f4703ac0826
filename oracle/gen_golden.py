"""Generate the golden vectors under tests/golden/ from the *reference itself*.

TEST INFRASTRUCTURE.  Runs only in the build container, where the reference is
mounted read-only at /root/reference (it does not exist on the GPU box; nothing
in tests/, smoke() or bench.py reads it at run time).  The reference is imported
as-is; four modules that are absent offline and only touched by off-path code
(HDF5 I/O, Brownian-tree samplers, CMS plot style, HGCal geometry unpickling) are
registered as empty stubs first.  Only *data* is written:
inputs, expected outputs, and small self-contained weight sets for the
per-primitive cases.  Full-model weights are never stored -- they are
re-created from the recorded torch seed by calodiffusion_amd.unet.CondUnet,
whose construction order makes its state_dict bit-identical to the reference's
(asserted here key by key), and verified through per-tensor fp64 checksums.

    python oracle/gen_golden.py            # writes tests/golden/*.npz
"""
from __future__ import annotations

import copy
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

for _name in ("h5py", "mplhep", "torchsde"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
_hg, _hgg = types.ModuleType("HGCalShowers"), types.ModuleType("HGCalShowers.HGCalGeo")
_hgg.HGCalGeo = type("HGCalGeo", (), {})
_hg.HGCalGeo = _hgg
sys.modules.setdefault("HGCalShowers", _hg)
sys.modules.setdefault("HGCalShowers.HGCalGeo", _hgg)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from calodiffusion.models import models as ref_models  # noqa: E402
from calodiffusion.models.calodiffusion import CaloDiffusion as RefCaloDiffusion  # noqa: E402
from calodiffusion.models import sample as ref_sample  # noqa: E402
from calodiffusion.utils import sampling as ref_sampling  # noqa: E402

from calodiffusion_amd import configs as my_configs  # noqa: E402
from calodiffusion_amd.unet import CondUnet as MyCondUnet, unet_kwargs_from_config  # noqa: E402

SEED = 1234
torch.set_num_threads(8)


def npf(t):
    return t.detach().cpu().numpy().astype(np.float32)


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def checksums(sd):
    """Per-tensor fp64 (sum, sum of squares) -- enough to catch any RNG/init drift."""
    keys = sorted(sd.keys())
    vals = np.array([[float(sd[k].double().sum()), float((sd[k].double() ** 2).sum())] for k in keys])
    return np.array(keys), vals


def build_ref(cfg):
    torch.manual_seed(SEED)
    m = RefCaloDiffusion(copy.deepcopy(cfg), n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
    m.eval()
    # our own parameter container must reproduce the reference init bit for bit
    torch.manual_seed(SEED)
    mine = MyCondUnet(**unet_kwargs_from_config(cfg))
    rsd, msd = m.model.state_dict(), mine.state_dict()
    assert list(rsd.keys()) == list(msd.keys()), "state_dict key order differs"
    for k in rsd:
        assert torch.equal(rsd[k], msd[k]), f"init mismatch at {k}"
    return m


def synth_inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    shape = [B] + list(cfg["SHAPE_PAD"][1:])
    x = torch.randn(shape, generator=g)
    n_e = 3 if cfg.get("HGCAL", False) else 1
    E = torch.rand((B, n_e), generator=g)
    layers = None
    if "layer" in cfg.get("SHOWERMAP", ""):
        layers = torch.randn((B, 1 + cfg["SHAPE_FINAL"][2]), generator=g)
    return x, E, layers


# ------------------------------------------------------------------ 1. known answer
def gold_cyl_known_answer():
    # calodiffusion/tests/test_cyl_conv.py: x = [[1,2,3]]*4 as (1,1,1,4,3), all-ones 1x3x3 kernel
    x = torch.tensor([[[[[1.0, 2, 3]] * 4]]])
    cyl = ref_models.CylindricalConv(1, 1, kernel_size=(1, 3, 3), stride=1, padding=(0, 1, 1), bias=False)
    plain = torch.nn.Conv3d(1, 1, kernel_size=(1, 3, 3), stride=1, padding=(0, 1, 1), bias=False)
    with torch.no_grad():
        cyl.conv.weight.fill_(1.0)
        plain.weight.fill_(1.0)
        save("cyl_known_answer", x=npf(x), cyl=npf(cyl(x)), plain=npf(plain(x)))


# ------------------------------------------------------------------ 2. primitives
def gold_primitives():
    torch.manual_seed(SEED + 1)
    out = {}
    with torch.no_grad():
        # 3x3x3 cylindrical convs on odd-sized grids
        for tag, (ci, co, shp) in {
            "c3_4_32": (4, 32, (2, 4, 5, 6, 7)),
            "c3_3_32": (3, 32, (1, 3, 9, 16, 9)),
            "c3_32_32": (32, 32, (2, 32, 5, 6, 4)),
            "c3_64_32": (64, 32, (1, 64, 4, 3, 5)),
            "c3_96_64": (96, 64, (1, 96, 3, 5, 2)),
        }.items():
            m = ref_models.CylindricalConv(ci, co, kernel_size=3, padding=1)
            x = torch.randn(shp)
            out.update({f"{tag}.x": npf(x), f"{tag}.w": npf(m.conv.weight), f"{tag}.b": npf(m.conv.bias),
                        f"{tag}.y": npf(m(x))})
        # 1x1x1 convs (with / without bias)
        for tag, (ci, co, bias) in {"c1_64_32": (64, 32, True), "c1_32_96": (32, 96, False), "c1_32_1": (32, 1, True)}.items():
            m = ref_models.CylindricalConv(ci, co, kernel_size=1, bias=bias)
            x = torch.randn((2, ci, 3, 4, 5))
            out.update({f"{tag}.x": npf(x), f"{tag}.w": npf(m.conv.weight), f"{tag}.y": npf(m(x))})
            if bias:
                out[f"{tag}.b"] = npf(m.conv.bias)
        # Downsample: compress_Z True / False, odd and even extents
        for tag, (c, shp, cz) in {
            "down_d2": (32, (1, 32, 9, 8, 9), True),
            "down_odd": (32, (2, 32, 5, 5, 7), True),
            "down_noz": (32, (1, 32, 4, 6, 4), False),
        }.items():
            m = ref_models.Downsample(c, cylindrical=True, compress_Z=cz)
            x = torch.randn(shp)
            out.update({f"{tag}.x": npf(x), f"{tag}.w": npf(m.conv.weight), f"{tag}.b": npf(m.conv.bias),
                        f"{tag}.y": npf(m(x)), f"{tag}.cz": np.array(int(cz))})
        # Upsample variants: D2 k=(3,4,4) op (0,0,0)/(0,0,1); HGCal k=(4,4,4); D1 op (0,1,1); no-z-compress
        for tag, (c, shp, extra, cz) in {
            "up_k3_op000": (32, (1, 32, 4, 4, 2), [0, 0, 0], True),
            "up_k3_op001": (32, (2, 32, 5, 4, 4), [0, 0, 1], True),
            "up_k4_op000": (32, (1, 32, 3, 3, 5), [1, 0, 0], True),
            "up_k4_op001": (64, (1, 64, 4, 6, 3), [1, 0, 1], True),
            "up_k3_op011": (32, (1, 32, 2, 2, 7), [0, 1, 1], True),
            "up_noz": (32, (1, 32, 4, 3, 2), [0, 0, 0], False),
        }.items():
            m = ref_models.Upsample(c, list(extra), cylindrical=True, compress_Z=cz)
            x = torch.randn(shp)
            out.update({f"{tag}.x": npf(x), f"{tag}.w": npf(m.convTrans.weight), f"{tag}.b": npf(m.convTrans.bias),
                        f"{tag}.y": npf(m(x)), f"{tag}.extra": np.array(extra), f"{tag}.cz": np.array(int(cz))})
    save("primitives_conv", **out)

    out = {}
    torch.manual_seed(SEED + 2)
    with torch.no_grad():
        # ResnetBlock (with cond MLP, with and without res_conv) and cond-less final block
        for tag, (ci, co, cond, shp) in {
            "res_32_32": (32, 32, 128, (2, 32, 5, 6, 4)),
            "res_32_64": (32, 64, 128, (2, 32, 4, 4, 3)),
            "res_128_32": (128, 32, 128, (1, 128, 3, 4, 2)),
            "res_nocond": (32, 32, None, (1, 32, 4, 4, 4)),
        }.items():
            m = ref_models.ResnetBlock(ci, co, cond_emb_dim=cond, groups=8, cylindrical=True)
            # non-trivial affine parameters for the norms
            for n, p in m.named_parameters():
                if "norm" in n:
                    p.copy_(torch.randn_like(p) * 0.3 + (1.0 if n.endswith("weight") else 0.0))
            x = torch.randn(shp)
            c = torch.randn((shp[0], cond)) if cond else None
            y = m(x, c)
            out[f"{tag}.x"], out[f"{tag}.y"] = npf(x), npf(y)
            if cond:
                out[f"{tag}.cond"] = npf(c)
            for k, v in m.state_dict().items():
                out[f"{tag}.sd.{k}"] = npf(v)
        # Residual(PreNorm(LinearAttention))
        for tag, (c, shp) in {"attn_32": (32, (2, 32, 5, 6, 4)), "attn_64": (64, (1, 64, 3, 4, 5)),
                              "attn_96": (96, (1, 96, 2, 3, 5))}.items():
            m = ref_models.Residual(ref_models.PreNorm(c, ref_models.LinearAttention(c, cylindrical=True)))
            for n, p in m.named_parameters():
                if "norm" in n or "to_out.1" in n:
                    p.copy_(torch.randn_like(p) * 0.3 + (1.0 if n.endswith("weight") else 0.0))
            x = torch.randn(shp) * 2.0
            out[f"{tag}.x"], out[f"{tag}.y"] = npf(x), npf(m(x))
            for k, v in m.state_dict().items():
                out[f"{tag}.sd.{k}"] = npf(v)
    save("primitives_blocks", **out)


# ------------------------------------------------------------------ 3. schedules
def gold_schedules():
    out = {}
    for n in (2, 10, 50, 200, 400):
        betas = ref_sampling.cosine_beta_schedule(n)
        ac = torch.cumprod(1.0 - betas, axis=0)
        out[f"betas_{n}"] = npf(betas)
        out[f"alphas_cumprod_{n}"] = npf(ac)
    save("schedules", **out)


# ------------------------------------------------------------------ 4. full models
def model_case(tag, cfg, B, sigmas, with_unet_fwd=True):
    m = build_ref(cfg)
    keys, cks = checksums(m.model.state_dict())
    x, E, layers = synth_inputs(cfg, B, SEED + 10)
    out = {"seed": np.array(SEED), "ck_keys": keys, "ck_vals": cks, "x": npf(x), "E": npf(E)}
    if layers is not None:
        out["layers"] = npf(layers)
    with torch.no_grad():
        for i, s in enumerate(sigmas):
            sig = torch.full((B, 1, 1, 1, 1), s, dtype=torch.float32)
            xin = x * float(np.sqrt(1.0 + s * s))  # realistic magnitude: x_t ~ sqrt(sigma_d^2 + sigma^2)
            out[f"sigma_{i}"] = np.array(s, dtype=np.float32)
            out[f"denoise_{i}"] = npf(m.denoise(xin, E=E, sigma=sig, layers=layers))
    save(f"model_{tag}", **out)
    return m, (x, E, layers)


def gold_models():
    cfg2 = my_configs.load_config("dataset2")
    m2, (x, E, layers) = model_case("dataset2", cfg2, 2, [2.57e4, 1.02, 1.06e-2])

    # DDIM trajectories on Dataset-2
    out = {"start": npf(x), "E": npf(E), "layers": npf(layers)}
    ddim = ref_sample.DDim(cfg2)
    for n in (2, 10, 50, 400):
        xf, xs, x0s = ddim(m2, x, E, layers, n, 0, False)
        out[f"ddim_{n}"] = npf(xf)
        if n == 10:
            out["ddim_10_xs"] = np.stack([npf(t) for t in xs])
            out["ddim_10_x0s"] = np.stack([npf(t) for t in x0s])
        print("ddim", n, float(xf.abs().mean()))
    xf, _, _ = ddim(m2, x, E, layers, 10, 3, False)
    out["ddim_10_off3"] = npf(xf)
    save("ddim_dataset2", **out)

    # hybrid_weight / l2 loss value (training forward)
    g = torch.Generator().manual_seed(SEED + 20)
    noise = torch.randn(x.shape, generator=g)
    rnd = torch.randn((x.shape[0],), generator=g)
    with torch.no_grad():
        loss = m2.compute_loss(x, E, noise=noise, layers=layers, rnd_normal=rnd)
    save("loss_dataset2", data=npf(x), E=npf(E), layers=npf(layers), noise=npf(noise), rnd_normal=npf(rnd),
         loss=np.array(float(loss), dtype=np.float64))
    del m2

    cfg3 = my_configs.load_config("dataset3")
    m3, (x3, E3, l3) = model_case("dataset3", cfg3, 1, [80.0, 0.9, 2.0e-2])
    g = torch.Generator().manual_seed(SEED + 21)
    noise = torch.randn(x3.shape, generator=g)
    tt = torch.tensor([137])
    with torch.no_grad():
        loss = m3.compute_loss(x3, E3, noise=noise, layers=l3, time=tt)
    # NB: reference Diffusion.compute_loss drops `time` (diffusion.py:106-110) -> it is re-drawn from the
    # global RNG inside Loss.__call__; pin it by seeding right before and recording the draw.
    torch.manual_seed(99)
    t_draw = torch.randint(0, 400, (1,)).long()
    torch.manual_seed(99)
    with torch.no_grad():
        loss = m3.compute_loss(x3, E3, noise=noise, layers=l3)
    save("loss_dataset3", data=npf(x3), E=npf(E3), noise=npf(noise), time=t_draw.numpy(),
         loss=np.array(float(loss), dtype=np.float64))
    del m3

    cfgh = my_configs.load_config("hgcal")
    model_case("hgcal", cfgh, 2, [300.0, 1.5, 5.0e-2])

    # tiny 8x8x8 config (BASELINE configs[0]): DDPM, 50 steps, batch 4, per-step noise from a seeded stream
    cfgt = my_configs.load_config("tiny")
    mt, (xt, Et, lt) = model_case("tiny", cfgt, 4, [40.0, 1.0, 3.0e-2])
    ddpm = ref_sample.DDPM(cfgt)
    torch.manual_seed(777)  # the sampler draws torch.randn(x.shape) once per step from the global stream
    xf, xs, x0s = ddpm(mt, xt, Et, lt, 50, 0, False)
    save("ddpm_tiny", start=npf(xt), E=npf(Et), layers=npf(lt), noise_seed=np.array(777), ddpm_50=npf(xf),
         x_step10=npf(xs[10]), x0_step10=npf(x0s[10]))

    # Dataset-1 grid (5,10,30): the NN embedding needs an absent XML, so pin the bare U-Net on the regular grid
    torch.manual_seed(SEED)
    kw = dict(out_dim=1, layer_sizes=[32, 32, 64, 96], channels=4, cond_dim=128, resnet_block_groups=8, mid_attn=True,
              block_attn=True, compress_Z=True, cylindrical=True, data_shape=[1, 4, 5, 10, 30], time_embed=False,
              cond_embed=False, cond_size=7)
    u = ref_models.CondUnet(**copy.deepcopy(kw)).eval()
    torch.manual_seed(SEED)
    mine = MyCondUnet(**copy.deepcopy(kw))
    for k, v in u.state_dict().items():
        assert torch.equal(v, mine.state_dict()[k]), k
    keys, cks = checksums(u.state_dict())
    g = torch.Generator().manual_seed(SEED + 30)
    xx = torch.randn((2, 4, 5, 10, 30), generator=g)
    cc = torch.randn((2, 7), generator=g)
    tt = torch.randn((2,), generator=g)
    with torch.no_grad():
        yy = u(xx, cond=cc, time=tt)
    save("unet_d1grid", seed=np.array(SEED), ck_keys=keys, ck_vals=cks, x=npf(xx), cond=npf(cc), time=npf(tt), y=npf(yy))


def gold_euler():
    """EDM Euler trajectories on Dataset-2 from the reference's own Euler sampler (default options: deterministic)."""
    cfg2 = my_configs.load_config("dataset2")
    m2 = build_ref(cfg2)
    x, E, layers = synth_inputs(cfg2, 2, SEED + 10)
    out = {"start": npf(x), "E": npf(E), "layers": npf(layers)}
    euler = ref_sample.Euler(cfg2)
    for n in (5, 18):
        xf, xs, x0s = euler(m2, x, E, layers, n, 0, False)
        out[f"euler_{n}"] = npf(xf)
        out[f"tsteps_{n}"] = npf(euler.setup(n, 0))
        if n == 5:
            out["euler_5_x0s"] = np.stack([npf(t) for t in x0s])
        print("euler", n, float(xf.abs().mean()))
    xf, _, _ = euler(m2, x, E, layers, 18, 2, False)
    out["euler_18_off2"] = npf(xf)
    save("euler_dataset2", **out)


def gold_reverse_norm():
    """utils.ReverseNormCaloChall of the reference on synthetic normalised showers (Dataset-2 with layer energies, Dataset-3)."""
    from calodiffusion.utils import utils as ref_utils
    g = torch.Generator().manual_seed(SEED + 40)
    out = {}
    for tag, dims, dnum, smap in (("d2", (45, 16, 9), 2, "layer-logit-norm"), ("d3", (6, 50, 18), 3, "logit-norm")):
        B = 3
        vox = (torch.randn((B, 1) + dims, generator=g) * 1.3 + 0.2).numpy().astype(np.float32)
        vox[0, 0, 3] = -30.0   # an (almost) empty layer: the rescale factor falls back to 1
        e = torch.rand((B, 1), generator=g).numpy().astype(np.float32)
        layerE = torch.randn((B, dims[0] + 1), generator=g).numpy().astype(np.float32) if "layer" in smap else None
        data, energy = ref_utils.ReverseNormCaloChall(vox.copy(), e.copy(), emax=1000., emin=1., max_deposit=2, logE=True,
                                                      layerE=None if layerE is None else layerE.copy(), showerMap=smap,
                                                      dataset_num=dnum, orig_shape=False, ecut=0.0000151)
        out[f"{tag}.vox"], out[f"{tag}.e"] = vox, e
        if layerE is not None:
            out[f"{tag}.layerE"] = layerE
        out[f"{tag}.data"], out[f"{tag}.energy"] = np.asarray(data, dtype=np.float32), np.asarray(energy, dtype=np.float32)
        print("reverse_norm", tag, float(np.abs(data).mean()))
    save("reverse_norm", **out)


class _ReshapeDecoder:
    """Stand-in for HGCalConverter (needs a geometry pickle that does not ship): dec_batches flattens (phi, r) into cells, so
    that the arithmetic AROUND the decode of utils.ReverseNormHGCal can be pinned."""

    def dec_batches(self, data, sparse_decoding=False, sparse_per_batch=False):
        d = np.squeeze(np.asarray(data), axis=1)
        return d.reshape(d.shape[0], d.shape[1], -1)


def gold_reverse_norm_hgcal():
    """utils.ReverseNormHGCal of the reference (utils/HGCal_utils.py:167-292) on synthetic normalised HGCal-shaped showers:
    dataset 111 (the shipped config) with layer energies, dataset 120 without, both through the stand-in decoder (the reference's
    final scaling only broadcasts for decoded (B, L, cells) showers)."""
    from calodiffusion.utils import HGCal_utils as ref_hg
    g = torch.Generator().manual_seed(SEED + 41)
    B, dims = 3, (28, 12, 21)
    vox = (torch.randn((B, 1) + dims, generator=g) * 0.9 + 0.3).numpy().astype(np.float32)
    vox[1, 0, 5] = -9.0  # an (almost) empty layer: the rescale factor falls back to 1
    e = torch.rand((B, 3), generator=g).numpy().astype(np.float32)
    layerE = torch.randn((B, dims[0] + 1), generator=g).numpy().astype(np.float32)
    out = {"vox": vox, "e": e, "layerE": layerE}
    data, gen = ref_hg.ReverseNormHGCal(vox.copy(), e.copy(), emax=1000., emin=1., max_deposit=2, logE=True, layerE=layerE.copy(),
                                        showerMap="layer-logit-norm", dataset_num=111, embed=True, NN_embed=_ReshapeDecoder())
    out["layer.data"], out["layer.gen"] = np.asarray(data, dtype=np.float32), np.asarray(gen, dtype=np.float32)
    data, gen = ref_hg.ReverseNormHGCal(vox.copy(), e.copy(), emax=1000., emin=1., max_deposit=2, logE=True, layerE=None,
                                        showerMap="logit-norm", dataset_num=120, embed=True, NN_embed=_ReshapeDecoder())
    out["plain.data"], out["plain.gen"] = np.asarray(data, dtype=np.float32), np.asarray(gen, dtype=np.float32)
    # dataset 121 in layer mode: layers_mean -11.6 / layers_std 7.3 put many layer energies near reverse_logit's alpha (1e-8
    # here, 1e-6 in utils.py), so this case separates the two
    data, gen = ref_hg.ReverseNormHGCal(vox.copy(), e.copy(), emax=1000., emin=1., max_deposit=2, logE=True, layerE=layerE.copy(),
                                        showerMap="layer-logit-norm", dataset_num=121, embed=True, NN_embed=_ReshapeDecoder())
    out["layer121.data"], out["layer121.gen"] = np.asarray(data, dtype=np.float32), np.asarray(gen, dtype=np.float32)
    print("reverse_norm hgcal", out["layer.data"].shape, float(np.abs(out["layer.data"]).mean()), out["plain.data"].shape)
    save("reverse_norm_hgcal", **out)


def gold_layer():
    """LayerDiffusion (models/layerdiffusion.py): the ResNet layer model's forward / denoise, sample_layers trajectories and one
    two-stage sample() on Dataset-2, from the reference's own classes.  Parameters come from torch.manual_seed(SEED)."""
    from calodiffusion.models.layerdiffusion import LayerDiffusion as RefLayerDiffusion
    from calodiffusion_amd.resnet import ResNet as MyResNet
    cfg = my_configs.load_config("dataset2")
    cfg["LAYER_STEPS"] = 12
    torch.manual_seed(SEED)
    m = RefLayerDiffusion(copy.deepcopy(cfg), n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
    m.eval()
    torch.manual_seed(SEED)
    mine_l = MyResNet(dim_in=cfg["SHAPE_FINAL"][2] + 1, num_layers=5, cond_size=1)
    mine_u = MyCondUnet(**unet_kwargs_from_config(cfg))
    for ref_mod, my_mod in ((m.layer_model, mine_l), (m.base_model, mine_u)):
        rsd, msd = ref_mod.state_dict(), my_mod.state_dict()
        assert list(rsd.keys()) == list(msd.keys()), "state_dict key order differs"
        for k in rsd:
            assert torch.equal(rsd[k], msd[k]), f"init mismatch at {k}"
    keys, cks = checksums(m.layer_model.state_dict())
    ukeys, ucks = checksums(m.base_model.state_dict())
    B, dim = 3, cfg["SHAPE_FINAL"][2] + 1
    g = torch.Generator().manual_seed(SEED + 50)
    x = torch.randn((B, dim), generator=g)
    E = torch.rand((B, 1), generator=g)
    t = torch.randn((B,), generator=g)
    start = torch.randn((B, dim), generator=g)
    shower_start = torch.randn([B] + list(cfg["SHAPE_PAD"][1:]), generator=g)
    out = {"seed": np.array(SEED), "ck_keys": keys, "ck_vals": cks, "unet_ck_keys": ukeys, "unet_ck_vals": ucks, "x": npf(x),
           "E": npf(E), "time": npf(t), "start": npf(start), "shower_start": npf(shower_start)}
    with torch.no_grad():
        out["forward"] = npf(m.layer_model(x, cond=E, time=t))
        m.set_layer_state(is_layer=True)
        for i, s in enumerate((80.0, 1.3, 2.0e-2)):
            sig = torch.full((B, 1), s)
            out[f"sigma_{i}"] = np.array(s, dtype=np.float32)
            out[f"denoise_{i}"] = npf(m.denoise(x * float(np.sqrt(0.25 + s * s)), E=E, sigma=sig, layers=None))
        m.set_layer_state(is_layer=False)
        draws = []
        m.noise_generation = lambda shape: draws.pop(0)
        for n in (12, 400):
            m.layer_steps = n
            draws[:] = [start]
            out[f"layers_{n}"] = npf(m.sample_layers(E, layers=None, sample_offset=0))
            print("layer ddim", n, float(np.abs(out[f"layers_{n}"]).mean()))
        m.layer_steps = 12
        draws[:] = [start]
        out["layers_12_off2"] = npf(m.sample_layers(E, layers=None, sample_offset=2))
        m.layer_sampler = ref_sample.Euler(cfg)
        draws[:] = [start]
        out["layers_euler_12"] = npf(m.sample_layers(E, layers=None, sample_offset=0))
        m.layer_sampler = ref_sample.DDim(cfg)
        draws[:] = [shower_start, start]  # sample() draws the shower start first, then the layer start
        res = m.sample(E, layers=None, num_steps=3, sample_offset=0, return_layers=True)
        out["sample_3_x"], out["sample_3_layers"] = np.asarray(res["x"], dtype=np.float32), npf(res["layers"])
        print("two-stage sample", float(np.abs(out["sample_3_x"]).mean()))
        # layer-state training loss (layerdiffusion.py:52-57): the noise it draws is pinned through noise_generation
        lay = torch.randn((B, dim), generator=g)
        lnoise = torch.randn((B, dim), generator=g)
        lrnd = torch.randn((B,), generator=g)
        m.set_layer_state(is_layer=True)
        draws[:] = [lnoise]
        out["loss"] = np.array(float(m.compute_loss(None, E, None, lay, rnd_normal=lrnd)), dtype=np.float64)
        m.set_layer_state(is_layer=False)
        out["loss_layers"], out["loss_noise"], out["loss_rnd"] = npf(lay), npf(lnoise), npf(lrnd)
        print("layer loss", float(out["loss"]))
    save("layer_dataset2", **out)


class count_draws:
    """Counts the torch.randn / torch.randn_like calls a reference sampler makes (the tests replay the same sequence of draws
    from the recorded seed: the CPU generator is deterministic for a given torch version)."""

    def __enter__(self):
        self.n, self.shapes = 0, []
        self._rl, self._r = torch.randn_like, torch.randn

        def randn_like(t, *a, **k):
            self.n += 1
            self.shapes.append(tuple(t.shape))
            return self._rl(t, *a, **k)

        def randn(*a, **k):
            self.n += 1
            out = self._r(*a, **k)
            self.shapes.append(tuple(out.shape))
            return out

        torch.randn_like, torch.randn = randn_like, randn
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.randn = self._rl, self._r


def gold_samplers():
    """Trajectories of the reference's other samplers (models/sample.py: EDM Euler with churn / Heun / DPM2 / LMS / Restart,
    DPM / DPM++2S / DPM++2M, Consistency) on the tiny config (batch 3), plus one Heun case on Dataset-2.  Stochastic samplers
    draw from the global torch generator, seeded right before the call; the number and order of draws is recorded."""
    cfgt = my_configs.load_config("tiny")
    mt = build_ref(cfgt)
    x, E, layers = synth_inputs(cfgt, 3, SEED + 60)
    out = {"start": npf(x), "E": npf(E), "layers": npf(layers)}

    def run(tag, cls, n, cfg_over=None, opts=None, offset=0, keep=("x",), rows=3):
        cfg = copy.deepcopy(cfgt)
        cfg.update(cfg_over or {})
        if opts:
            cfg["SAMPLER_OPTIONS"] = opts
        smp = cls(cfg)
        seed = 4000 + len(out)
        torch.manual_seed(seed)
        with count_draws() as cd:
            with torch.no_grad():
                xf, xs, x0s = smp(mt, x[:rows].clone(), E[:rows], layers[:rows], n, offset, False)
        mt.loss_function.update_step(cfgt["NSTEPS"])  # DPM.setup / Consistency change the model's tables
        out[f"{tag}.n"], out[f"{tag}.seed"], out[f"{tag}.draws"] = np.array(n), np.array(seed), np.array(cd.n)
        out[f"{tag}.x"] = npf(xf)
        for k in keep:
            if k.startswith("xs"):
                out[f"{tag}.{k}"] = npf(xs[int(k[2:])])
            elif k.startswith("x0s"):
                out[f"{tag}.{k}"] = npf(x0s[int(k[3:])])
        fin = bool(torch.isfinite(xf).all())
        print(f"sampler {tag}: n={n} draws={cd.n} finite={fin} mean|x|={float(xf.abs().mean()) if fin else float('nan'):.4f}")
        return smp

    S = ref_sample
    run("euler_noisy", S.Euler, 8, {"NOISY_SAMPLE": True}, keep=("xs3", "x0s3", "xs7", "x0s7"))
    run("heun", S.Heun, 6, keep=("xs3", "x0s3", "xs5", "x0s5"))
    run("heun_noisy", S.Heun, 5, {"NOISY_SAMPLE": True}, keep=("xs4", "x0s4"))
    run("dpm2", S.DPM2, 6, keep=("xs3", "x0s3", "xs5", "x0s5"))
    run("dpm2_off1", S.DPM2, 6, offset=1, keep=("xs4",))
    run("lms", S.LMS, 9)
    run("lms_o2", S.LMS, 6, opts={"ORDER": 2})
    run("restart_default", S.Restart, 6, keep=("x0s5",))
    smp = S.Euler(copy.deepcopy(cfgt))
    tk = smp.setup(6, 0)
    rl = {2: [3, 1, 0.0, float(tk[2]) * 3.0], 4: [4, 2, 0.0, float(tk[4]) * 5.0]}
    run("restart_int", S.Restart, 6, opts={"RESTART_LIST": rl}, keep=("x0s5",))
    out["restart_int.keys"] = np.array(sorted(rl))
    out["restart_int.vals"] = np.array([rl[k] for k in sorted(rl)], dtype=np.float64)
    run("restart_noisy", S.Restart, 5, {"NOISY_SAMPLE": True}, opts={"RESTART_LIST": {3: [3, 1, 0.0, float(smp.setup(5, 0)[3]) * 4.0]}})
    out["restart_noisy.tmax"] = np.array(float(smp.setup(5, 0)[3]) * 4.0)
    # The DPM family hands the model a (B,)-shaped sigma, which the reference's denoise multiplies into the (B,1,D,H,W) state
    # (calodiffusion.py:159): that broadcasts only for B = 1 (or, wrongly, B = W).  Batch 1 therefore.
    run("dpmpp2m", S.DPMPP2M, 9, rows=1)
    run("dpmpp2s", S.DPMPP2S, 5, rows=1)
    run("dpmpp2s_eta", S.DPMPP2S, 4, opts={"ETA": 1.0}, rows=1)
    run("dpm_7", S.DPM, 7, rows=1)
    run("dpm_6", S.DPM, 6, rows=1)
    run("dpm_2", S.DPM, 2, rows=1)
    run("consistency", S.Consistency, 3, {"CONSIS_NSTEPS": 40}, keep=("xs1",))
    # Consistency returns (x, xs, x0): the last denoised tensor
    save("samplers_tiny", **out)

    cfg2 = my_configs.load_config("dataset2")
    m2 = build_ref(cfg2)
    x2, E2, l2 = synth_inputs(cfg2, 1, SEED + 61)
    with torch.no_grad():
        xf, xs, x0s = S.Heun(copy.deepcopy(cfg2))(m2, x2.clone(), E2, l2, 4, 0, False)
        lf, _, _ = S.LMS(copy.deepcopy(cfg2))(m2, x2.clone(), E2, l2, 6, 0, False)
    save("samplers_dataset2", start=npf(x2), E=npf(E2), layers=npf(l2), heun_xs3=npf(xs[3]), heun_x0s3=npf(x0s[3]), lms_6=npf(lf))


def gold_sinusoidal():
    """CondUnet with sinusoidal time / cond embeddings (models.py:132-144, 578-601), called directly: CaloDiffusion cannot reach
    this branch (do_time_embed raises KeyError for 'sin', calodiffusion.py:148-152)."""
    out = {"seed": np.array(SEED)}
    for tag, kw_over, cond_shape in (("both", dict(time_embed=True, cond_embed=True, cond_size=1), (3,)),
                                     ("time", dict(time_embed=True, cond_embed=False, cond_size=10), (3, 10)),
                                     ("cond", dict(time_embed=False, cond_embed=True, cond_size=1), (3,))):
        kw = dict(out_dim=1, layer_sizes=[32, 32, 64, 32], channels=4, cond_dim=128, resnet_block_groups=8, mid_attn=True,
                  block_attn=True, compress_Z=True, cylindrical=True, data_shape=[1, 4, 8, 8, 8])
        kw.update(kw_over)
        torch.manual_seed(SEED)
        u = ref_models.CondUnet(**copy.deepcopy(kw)).eval()
        torch.manual_seed(SEED)
        mine = MyCondUnet(**copy.deepcopy(kw))
        rsd, msd = u.state_dict(), mine.state_dict()
        assert list(rsd.keys()) == list(msd.keys()), (list(rsd.keys())[:8], list(msd.keys())[:8])
        for k, v in rsd.items():
            assert torch.equal(v, msd[k]), k
        keys, cks = checksums(rsd)
        g = torch.Generator().manual_seed(SEED + 70)
        xx = torch.randn((3, 4, 8, 8, 8), generator=g)
        cc = torch.rand(cond_shape, generator=g) * 3.0
        tt = torch.rand((3,), generator=g) * 5.0 - 1.0
        with torch.no_grad():
            yy = u(xx, cond=cc, time=tt)
        out.update({f"{tag}.ck_keys": keys, f"{tag}.ck_vals": cks, f"{tag}.x": npf(xx), f"{tag}.cond": npf(cc), f"{tag}.time": npf(tt),
                    f"{tag}.y": npf(yy)})
        print("sinusoidal", tag, float(yy.abs().mean()))
    save("unet_sinusoidal", **out)


def gold_grads():
    """.grad of the reference's own compute_loss(...).backward() (models/loss.py:163-179, train/train_diffusion.py:52-63) on the
    inputs of loss_dataset2 / loss_dataset3: a few whole tensors and fp64 (sum, sum of squares) of every parameter's gradient."""
    pick = {"dataset2": ["init_conv.conv.weight", "downs.0.0.block1.proj.conv.weight", "downs_attn.1.fn.fn.to_qkv.conv.weight",
                         "ups.1.2.convTrans.weight", "time_mlp.1.weight", "cond_mlp.4.bias", "final_conv.1.conv.weight",
                         "mid_block1.block2.norm.weight"],
            "dataset3": ["init_conv.conv.weight", "downs.1.2.conv.weight", "ups_attn.0.fn.fn.to_out.0.conv.weight",
                         "ups.2.1.block2.proj.conv.weight", "cond_mlp.0.weight", "final_conv.0.block1.norm.bias"]}
    for name in ("dataset2", "dataset3"):
        cfg = my_configs.load_config(name)
        m = build_ref(cfg)
        m.train()
        g = np.load(os.path.join(GOLD, f"loss_{name}.npz"))
        data, E, noise = (torch.from_numpy(g[k]) for k in ("data", "E", "noise"))
        layers = torch.from_numpy(g["layers"]) if "layers" in g else None
        m.zero_grad()
        if name == "dataset2":
            loss = m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=torch.from_numpy(g["rnd_normal"]))
        else:
            torch.manual_seed(99)  # the reference re-draws `time` inside Loss.__call__ (see gold_models)
            loss = m.compute_loss(data, E, noise=noise, layers=layers)
        assert abs(float(loss) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"])), (float(loss), float(g["loss"]))
        loss.backward()
        grads = {k: p.grad for k, p in m.model.named_parameters()}
        keys, cks = checksums(grads)
        out = {"ck_keys": keys, "ck_vals": cks, "loss": np.array(float(loss), dtype=np.float64)}
        for k in pick[name]:
            out["grad." + k] = npf(grads[k])
        save(f"grads_{name}", **out)
        print("grads", name, float(loss), float(sum(float((v.double() ** 2).sum()) for v in grads.values()) ** 0.5))


def gold_grads_round3():
    """Round 3: (a) HGCal .grad tensors of the reference's compute_loss(...).backward() (the round-2 fixtures cover Dataset-2 and
    Dataset-3 only), with their inputs; (b) the other LOSS_TYPEs of Loss._loss (models/loss.py:97-116: 'l1', 'mse', 'huber' --
    the reference's CI fixture trains with 'huber', tests/test_execution.py:94) on the tiny config: loss value and gradients."""
    cfg = my_configs.load_config("hgcal")
    m = build_ref(cfg)
    m.train()
    data, E, layers = synth_inputs(cfg, 2, SEED + 90)
    g = torch.Generator().manual_seed(SEED + 91)
    noise = torch.randn(data.shape, generator=g)
    rnd = torch.randn((2,), generator=g)
    m.zero_grad()
    loss = m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)
    loss.backward()
    grads = {k: p.grad for k, p in m.model.named_parameters()}
    keys, cks = checksums(grads)
    out = {"ck_keys": keys, "ck_vals": cks, "loss": np.array(float(loss), dtype=np.float64), "data": npf(data), "E": npf(E),
           "layers": npf(layers), "noise": npf(noise), "rnd_normal": npf(rnd)}
    for k in ("init_conv.conv.weight", "mid_attn.fn.fn.to_qkv.conv.weight", "downs.0.2.conv.weight",
              "ups.2.0.block1.proj.conv.weight", "ups_attn.0.fn.fn.to_out.0.conv.weight", "cond_mlp.0.weight",
              "final_conv.0.block2.norm.weight"):
        out["grad." + k] = npf(grads[k])
    save("grads_hgcal", **out)
    print("grads hgcal", float(loss), float(sum(float((v.double() ** 2).sum()) for v in grads.values()) ** 0.5))
    del m

    cfg = my_configs.load_config("tiny")
    data, E, layers = synth_inputs(cfg, 4, SEED + 92)
    g = torch.Generator().manual_seed(SEED + 93)
    noise = torch.randn(data.shape, generator=g)
    rnd = torch.randn((4,), generator=g)
    out = {"data": npf(data), "E": npf(E), "layers": npf(layers), "noise": npf(noise), "rnd_normal": npf(rnd)}
    for lt in ("l1", "mse", "huber", "l2"):
        c = copy.deepcopy(cfg)
        c["LOSS_TYPE"] = lt
        m = build_ref(c)
        m.train()
        m.zero_grad()
        loss = m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)
        loss.backward()
        grads = {k: p.grad for k, p in m.model.named_parameters()}
        keys, cks = checksums(grads)
        out[f"{lt}.ck_keys"], out[f"{lt}.ck_vals"], out[f"{lt}.loss"] = keys, cks, np.array(float(loss), dtype=np.float64)
        for k in ("init_conv.conv.weight", "mid_block1.block1.proj.conv.weight", "final_conv.1.conv.weight", "time_mlp.1.weight"):
            out[f"{lt}.grad.{k}"] = npf(grads[k])
        # how the residuals of this case straddle the huber knee (|d| = 1): both branches must be exercised
        with torch.no_grad():
            sigma = (rnd * m.loss_function.P_std + m.loss_function.P_mean).exp().reshape(-1, 1, 1, 1, 1)
            d = m.denoise(data + sigma * noise, E=E, sigma=sigma, layers=layers) - data
        out[f"{lt}.frac_abs_d_below_1"] = np.array(float((d.abs() < 1).float().mean()))
        print("loss type", lt, float(loss), float((d.abs() < 1).float().mean()))
    save("losstypes_tiny", **out)


def gold_objectives():
    """Round 4: TRAINING_OBJ 'noise_pred' and 'mean_pred' on the tiny config, from the reference's own classes: denoise at three
    noise levels (the branches of CaloDiffusion.denoise, models/calodiffusion.py:161-165), the loss value and .grad of
    compute_loss(...).backward() (models/loss.py:181-210) for LOSS_TYPE l2 and huber, and a 6-step DDIM end point (the sampler
    runs on whatever denoise returns).  'minsnr' cannot be constructed by the reference (TypeError, recorded)."""
    base = my_configs.load_config("tiny")
    data, E, layers = synth_inputs(base, 4, SEED + 95)
    g = torch.Generator().manual_seed(SEED + 96)
    noise = torch.randn(data.shape, generator=g)
    rnd = torch.randn((4,), generator=g)
    out = {"data": npf(data), "E": npf(E), "layers": npf(layers), "noise": npf(noise), "rnd_normal": npf(rnd),
           "sigmas": np.array([40.0, 1.02, 0.05], dtype=np.float32)}
    for obj in ("noise_pred", "mean_pred"):
        for lt in ("l2", "huber"):
            c = copy.deepcopy(base)
            c["TRAINING_OBJ"], c["LOSS_TYPE"] = obj, lt
            m = build_ref(c)
            assert type(m.loss_function).__name__ == obj
            if lt == "l2":
                with torch.no_grad():
                    for i, sg in enumerate(out["sigmas"]):
                        sig = torch.full((4, 1, 1, 1, 1), float(sg))
                        xin = data * float(np.sqrt(1.0 + float(sg) ** 2))
                        out[f"{obj}.denoise_{i}"] = npf(m.denoise(xin, E=E, sigma=sig, layers=layers))
                    xf, _, _ = ref_sample.DDim(c)(m, data, E, layers, 6, 0, False)
                    out[f"{obj}.ddim_6"] = npf(xf)
            m.train()
            m.zero_grad()
            loss = m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)
            loss.backward()
            grads = {k: p.grad for k, p in m.model.named_parameters()}
            keys, cks = checksums(grads)
            out[f"{obj}.{lt}.ck_keys"], out[f"{obj}.{lt}.ck_vals"] = keys, cks
            out[f"{obj}.{lt}.loss"] = np.array(float(loss), dtype=np.float64)
            for k in ("init_conv.conv.weight", "mid_block1.block1.proj.conv.weight", "final_conv.1.conv.weight", "time_mlp.1.weight",
                      "downs_attn.0.fn.fn.to_qkv.conv.weight"):
                out[f"{obj}.{lt}.grad.{k}"] = npf(grads[k])
            print("objective", obj, lt, float(loss), float(sum(float((v.double() ** 2).sum()) for v in grads.values()) ** 0.5))
    c = copy.deepcopy(base)
    c["TRAINING_OBJ"] = "minsnr"
    try:
        build_ref(c)
        out["minsnr.constructs"] = np.array(1)
    except TypeError as e:
        out["minsnr.constructs"] = np.array(0)
        print("minsnr:", e)
    save("objectives_tiny", **out)


def gold_dpm_tables():
    """The DPM-Solver-fast step tables (calodiffusion_amd.sample.DPM.build) of the three `dpm_*` sampler cases AS COMPUTED ON THIS
    HOST, where tests/test_host.py shows the programs reproduce the reference's trajectories bit for bit.  torch's vectorised
    cos / exp / log / expm1 differ in the last bit between CPUs (measured: this container's Xeon against the GPU box's EPYC 9575F:
    the 2-step table's entries differ by 1-5 ulp), and `dpm_2` amplifies a 4e-7 change of its coefficients to 1.5e-4 of its end
    point -- the reference itself, run on that other host, lands 1.5e-4 from its own result here.  With these tables the
    device test can hold the reference's trajectory from THIS host on any host."""
    from sampler_cases import CASES
    from calodiffusion_amd.calodiffusion import CaloDiffusion as MyCaloDiffusion
    g = np.load(os.path.join(GOLD, "samplers_tiny.npz"))
    out = {}
    for tag in ("dpm_7", "dpm_6", "dpm_2"):
        name, over, _, off, rows = CASES[tag]
        cfg = copy.deepcopy(my_configs.load_config("tiny"))
        cfg.update(over)
        cfg["SAMPLER"] = name
        m = MyCaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
        prog = m.sampler_algorithm.build(m, int(g[f"{tag}.n"]), off).finalize()
        out[f"{tag}.coefs"], out[f"{tag}.start_scale"] = prog.coefs, np.array(prog.start_scale, dtype=np.float64)
    save("dpm_tables", **out)


def gold_trajectories():
    """Dataset-3 DDIM (10 and 50 steps, batch 1) and HGCal DDPM (200 steps, batch 2, seeded noise stream) end points."""
    cfg3 = my_configs.load_config("dataset3")
    m3 = build_ref(cfg3)
    x, E, layers = synth_inputs(cfg3, 1, SEED + 80)
    out = {"start": npf(x), "E": npf(E)}
    ddim = ref_sample.DDim(cfg3)
    with torch.no_grad():
        for n in (10, 50):
            xf, xs, x0s = ddim(m3, x, E, layers, n, 0, False)
            out[f"ddim_{n}"] = npf(xf)
            print("d3 ddim", n, float(xf.abs().mean()))
    save("ddim_dataset3", **out)
    del m3
    cfgh = my_configs.load_config("hgcal")
    mh = build_ref(cfgh)
    x, E, layers = synth_inputs(cfgh, 2, SEED + 81)
    ddpm = ref_sample.DDPM(cfgh)
    torch.manual_seed(778)
    with torch.no_grad():
        xf, xs, x0s = ddpm(mh, x, E, layers, 200, 0, False)
    print("hgcal ddpm 200", float(xf.abs().mean()))
    save("ddpm_hgcal", start=npf(x), E=npf(E), layers=npf(layers), noise_seed=np.array(778), ddpm_200=npf(xf),
         x_step100=npf(xs[100]), x0_step100=npf(x0s[100]))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["known", "prims", "sched", "models"]
    if "known" in which:
        gold_cyl_known_answer()
    if "prims" in which:
        gold_primitives()
    if "sched" in which:
        gold_schedules()
    if "models" in which:
        gold_models()
    if "euler" in which:
        gold_euler()
    if "renorm" in which:
        gold_reverse_norm()
    if "renorm_hgcal" in which:
        gold_reverse_norm_hgcal()
    if "layer" in which:
        gold_layer()
    if "samplers" in which:
        gold_samplers()
    if "sin" in which:
        gold_sinusoidal()
    if "grads" in which:
        gold_grads()
    if "dpm_tables" in which:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        gold_dpm_tables()
    if "objectives" in which:
        gold_objectives()
    if "traj" in which:
        gold_trajectories()
    if "grads3" in which:
        gold_grads_round3()

