"""GPU (MI355X): the deepest U-Net level as one launch (kernels_deep.hip: downs[-1], the mid blocks and ups[0] -- six ResnetBlocks,
three attention blocks -- with the activations resident in LDS), against the CPU oracle through CondUnet.forward on grids whose
deepest level is 8 .. 120 voxels (1 .. 4 row tiles, ragged and full), against the reference's goldens through denoise, and
against the per-op kernels it replaces (CD_NO_DEEP_LEVEL=1) on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import gold, rel_l2
from helpers import seeded_unet, t

pytestmark = pytest.mark.gpu

TOL_OP = 1e-5


def _unet_kwargs(grid, sizes=(32, 32, 64, 32), channels=3, cond_size=9, **over):
    kw = dict(out_dim=1, layer_sizes=list(sizes), channels=channels, cond_dim=128, resnet_block_groups=8, mid_attn=True,
              block_attn=True, compress_Z=True, cylindrical=True, data_shape=[1, channels] + list(grid), time_embed=False,
              cond_embed=False, cond_size=cond_size)
    kw.update(over)
    return kw


def _deep_level_launches(fn):
    """Run fn() under the per-launch profiler and return {category: launches}."""
    from calodiffusion_amd import engine
    engine.profile_begin()
    fn()
    return {k: v["launches"] for k, v in engine.profile_end().items()}


@pytest.mark.parametrize("grid,sizes,B", [
    ((8, 8, 8), (32, 32, 64, 32), 3),      # deepest level 2x2x2 = 8 voxels: one ragged row tile
    ((21, 12, 8), (32, 32, 64, 32), 2),    # 6x3x2 = 36 voxels: two tiles, the second ragged
    ((45, 16, 9), (32, 32, 64, 32), 2),    # Dataset-2's own level: 12x4x2 = 96 voxels, three full tiles, 64 <-> 32 channels
    ((45, 20, 8), (32, 32, 32, 32), 2),    # 12x5x2 = 120 voxels: four tiles, 32 channels throughout (no shortcut conv)
    ((23, 8, 4), (32, 32, 64), 2),         # two levels only: deepest 12x4x2 with 32 -> 64 -> 32 ... -> 32 channels (Cb = 64)
])
def test_deep_level_through_unet_forward_matches_oracle(grid, sizes, B):
    from oracle import torch_oracle as O
    kw = _unet_kwargs(grid, sizes)
    net = seeded_unet(kw, 77).cuda()
    g = torch.Generator().manual_seed(5)
    x = torch.randn([B, 3] + list(grid), generator=g)
    cond, time = torch.randn((B, 9), generator=g), torch.rand((B,), generator=g)
    spec = O.UnetSpec(layer_sizes=list(sizes), channels=3, cond_size=9, data_shape=tuple(grid))
    with torch.no_grad():
        want = O.cond_unet_forward({k: v.cpu() for k, v in net.state_dict().items()}, spec, x, cond, time)
    launches = _deep_level_launches(lambda: net(x.cuda(), cond=cond.cuda(), time=time.cuda()))
    assert any(k.startswith("deep_level") for k in launches), sorted(launches)
    got = net(x.cuda(), cond=cond.cuda(), time=time.cuda())
    err = rel_l2(got.cpu().numpy(), want.numpy())
    print(f"deep level, grid {grid} sizes {sizes}: unet_forward rel L2 {err:.2e}")
    assert err < TOL_OP


def test_deep_level_equals_the_per_op_kernels(monkeypatch):
    """Same model, same inputs: the one-launch level against the 13 launches it replaces (both fp32-grade, different summation
    orders); and the launch count of a denoise step."""
    from test_gpu_parity import _model
    m = _model("dataset2")
    cfg = m.config
    B = 5
    g = torch.Generator().manual_seed(17)
    x = torch.randn([B] + list(cfg["SHAPE_PAD"][1:]), generator=g).cuda()
    E, layers = torch.rand((B, 1), generator=g).cuda(), torch.randn((B, cfg["SHAPE_PAD"][2] + 1), generator=g).cuda()
    sig = torch.tensor([80.0, 5.0, 1.0, 0.3, 0.02]).cuda()
    fused = m.denoise(x, E=E, sigma=sig, layers=layers)
    n_fused = _deep_level_launches(lambda: m.denoise(x, E=E, sigma=sig, layers=layers))
    monkeypatch.setenv("CD_NO_DEEP_LEVEL", "1")
    per_op = m.denoise(x, E=E, sigma=sig, layers=layers)
    n_per_op = _deep_level_launches(lambda: m.denoise(x, E=E, sigma=sig, layers=layers))
    monkeypatch.delenv("CD_NO_DEEP_LEVEL")
    err = float((fused - per_op).norm() / per_op.norm())
    print(f"deep level vs per-op kernels: rel L2 {err:.2e}; launches per denoise {sum(n_fused.values())} vs {sum(n_per_op.values())}")
    assert err < 3e-6
    assert not any(k.startswith("deep_level") for k in n_per_op) and sum(n_fused.values()) <= sum(n_per_op.values()) - 10
    assert sum(n_fused.values()) <= 60, n_fused  # launches per denoise step <= 60
    # batch independence and determinism of the one-launch level
    again = m.denoise(x, E=E, sigma=sig, layers=layers)
    one = m.denoise(x[2:3], E=E[2:3], sigma=sig[2:3], layers=layers[2:3])
    assert torch.equal(again, fused)
    # (not bitwise: the level-0 kernels size their chunks -- and with them their summation order -- by the batch)
    assert float((one[0] - fused[2]).norm() / fused[2].norm()) < 1e-6


def test_deep_level_flags_fp16_overflow_like_the_convs_it_replaces():
    """An out-of-range activation inside the level raises the plan's range flag (and the safe denoise falls back)."""
    from test_gpu_parity import _model
    m = _model("dataset2")
    cfg = m.config
    x = torch.randn([1] + list(cfg["SHAPE_PAD"][1:]), device="cuda")
    E, layers = torch.rand((1, 1), device="cuda"), torch.randn((1, cfg["SHAPE_PAD"][2] + 1), device="cuda")
    sig = torch.ones(1, device="cuda")
    with torch.no_grad():
        m.model.downs[1][2].conv.bias.fill_(3.0e5)  # the strided conv INTO the deepest level: its output is the level's input
    eng = m.engine()
    eng.safe_denoise = False
    m.denoise(x, E=E, sigma=sig, layers=layers)
    with pytest.raises(FloatingPointError):
        eng.check_status()
    eng.safe_denoise = True
    before = getattr(eng, "range_fallbacks", 0)
    out = m.denoise(x, E=E, sigma=sig, layers=layers)
    assert torch.isfinite(out).all() and eng.range_fallbacks == before + 1
