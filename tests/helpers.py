"""Shared test helpers: seeded model construction + checksum verification against the golden fixtures."""
import numpy as np
import torch

from calodiffusion_amd.configs import load_config
from calodiffusion_amd.unet import CondUnet, unet_kwargs_from_config

SEED = 1234


def seeded_unet(cfg_name_or_kwargs, seed=SEED):
    """CondUnet whose parameters equal the reference's for torch.manual_seed(seed) (see oracle/gen_golden.py)."""
    kw = unet_kwargs_from_config(load_config(cfg_name_or_kwargs)) if isinstance(cfg_name_or_kwargs, str) else dict(cfg_name_or_kwargs)
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = CondUnet(**kw)
    torch.random.set_rng_state(state)
    return net


def verify_checksums(sd, g):
    keys = [str(k) for k in g["ck_keys"]]
    vals = g["ck_vals"]
    assert sorted(sd.keys()) == keys
    for k, (s1, s2) in zip(keys, vals):
        t = sd[k].double()
        assert abs(float(t.sum()) - s1) <= 1e-9 * max(1.0, abs(s1)), k
        assert abs(float((t * t).sum()) - s2) <= 1e-9 * max(1.0, abs(s2)), k


def d1grid_kwargs():
    return dict(out_dim=1, layer_sizes=[32, 32, 64, 96], channels=4, cond_dim=128, resnet_block_groups=8, mid_attn=True,
                block_attn=True, compress_Z=True, cylindrical=True, data_shape=[1, 4, 5, 10, 30], time_embed=False,
                cond_embed=False, cond_size=7)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def seeded_layer_models(cfg_name="dataset2", seed=SEED):
    """(ResNet layer model, CondUnet) with the parameters the reference's LayerDiffusion gets for torch.manual_seed(seed):
    the layer model is constructed first (layerdiffusion.py:35-40)."""
    from calodiffusion_amd.resnet import ResNet
    cfg = load_config(cfg_name)
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    layer = ResNet(dim_in=cfg["SHAPE_FINAL"][2] + 1, num_layers=5, cond_size=3 if cfg.get("HGCAL", False) else 1)
    unet = CondUnet(**unet_kwargs_from_config(cfg))
    torch.random.set_rng_state(state)
    return layer, unet
