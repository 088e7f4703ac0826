"""GPU (MI355X): LayerDiffusion's two-stage generation on the HIP library, through the C ABI, against the reference's outputs
(tests/golden/layer_dataset2.npz, made by oracle/gen_golden.py from reference models/layerdiffusion.py) and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import gold, rel_l2
from helpers import t, verify_checksums
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu

TOL_OP = 1e-5
TOL_TRAJ = 1e-4


def _model(**extra):
    from calodiffusion_amd.layerdiffusion import LayerDiffusion
    from calodiffusion_amd.configs import load_config
    cfg = load_config("dataset2")
    cfg["LAYER_STEPS"] = 12
    cfg.update(extra)
    torch.manual_seed(1234)
    return LayerDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])


def test_layer_model_forward_and_denoise():
    g = gold("layer_dataset2")
    m = _model()
    verify_checksums({k: v.cpu() for k, v in m.layer_model.state_dict().items()}, g)
    x, E, tm = t(g["x"]).cuda(), t(g["E"]).cuda(), t(g["time"]).cuda()
    assert rel_l2(m.layer_model(x, cond=E, time=tm).cpu().numpy(), g["forward"]) < TOL_OP
    m.set_layer_state(is_layer=True)
    assert rel_l2(m.forward(x, E, tm, layers=None).cpu().numpy(), g["forward"]) < TOL_OP
    for i in range(3):
        s = float(g[f"sigma_{i}"])
        y = m.denoise(x * float(np.sqrt(0.25 + s * s)), E=E, sigma=torch.full((3, 1), s, device="cuda"), layers=None)
        assert rel_l2(y.cpu().numpy(), g[f"denoise_{i}"]) < TOL_OP, i
    m.set_layer_state(is_layer=False)
    with pytest.raises(ValueError):
        m.layer_model(x[:, :40].contiguous(), cond=E, time=tm)


def test_sample_layers_trajectories():
    """sample_layers (layerdiffusion.py:114-132): one launch per trajectory; DDim at 12 and 400 steps, with a sample offset,
    and the EDM Euler sampler, against the reference; per-step records against the oracle."""
    from calodiffusion_amd import sample
    g = gold("layer_dataset2")
    m = _model()
    E, start = t(g["E"]).cuda(), t(g["start"]).cuda()
    for n in (12, 400):
        m.layer_steps = n
        y = m.sample_layers(E, layers=None, sample_offset=0, start=start)
        assert rel_l2(y.cpu().numpy(), g[f"layers_{n}"]) < TOL_TRAJ, n
    m.layer_steps = 12
    y = m.sample_layers(E, layers=None, sample_offset=2, start=start)
    assert rel_l2(y.cpu().numpy(), g["layers_12_off2"]) < TOL_TRAJ
    assert not m.layer_loss and m.model is m.base_model
    m.layer_sampler = sample.Euler(m.config)
    y = m.sample_layers(E, layers=None, sample_offset=0, start=start)
    assert rel_l2(y.cpu().numpy(), g["layers_euler_12"]) < TOL_TRAJ
    # trajectories and the stochastic sampler (noise handed in) against the oracle
    om = O.OracleLayerModel(m.config, {k: v.cpu() for k, v in m.layer_model.state_dict().items()})
    gen = torch.Generator().manual_seed(5)
    noise = torch.randn((12,) + tuple(start.shape), generator=gen)
    ddpm = sample.DDPM(m.config)
    ddpm.step_noise = noise.cuda()
    m.set_layer_state(is_layer=True)
    x, xs, x0s = ddpm(m, start, E, None, 12, 0, True)
    m.set_layer_state(is_layer=False)
    with torch.no_grad():
        wx, wxs, wx0s = om.ddim_sample(start.cpu(), E.cpu(), None, 12, eta=1.0, step_noise=list(noise), keep=True)
    assert rel_l2(x.cpu().numpy(), wx.numpy()) < TOL_TRAJ
    assert rel_l2(torch.stack(xs).cpu().numpy(), torch.stack(wxs).numpy()) < TOL_TRAJ
    assert rel_l2(torch.stack(x0s).cpu().numpy(), torch.stack(wx0s).numpy()) < TOL_TRAJ


def test_two_stage_sample_and_generate():
    """LayerDiffusion.sample / generate (layerdiffusion.py:134-235): generated layer energies condition the U-Net sampler."""
    g = gold("layer_dataset2")
    m = _model()
    verify_checksums({k: v.cpu() for k, v in m.base_model.state_dict().items()},
                     {"ck_keys": g["unet_ck_keys"], "ck_vals": g["unet_ck_vals"]})
    E = t(g["E"]).cuda()
    out = m.sample(E, layers=None, num_steps=3, sample_offset=0, return_layers=True, start=t(g["shower_start"]).cuda(),
                   layer_start=t(g["start"]).cuda())
    assert rel_l2(out["layers"].cpu().numpy(), g["sample_3_layers"]) < TOL_TRAJ
    assert rel_l2(out["x"], g["sample_3_x"]) < TOL_TRAJ
    # generation loop with device noise: physical showers of the reference's shapes, reproducible from the Philox offset
    loader = [(torch.rand((2, 1)), None, None) for _ in range(2)]
    m.noise_offset = 0
    a, e = m.generate(loader, sample_steps=3)
    m.noise_offset = 0
    b, _ = m.generate(loader, sample_steps=3)
    assert a.shape == (4, 6480) and e.shape == (4, 1) and np.isfinite(a).all() and (a >= 0).all()
    assert np.array_equal(a, b)
    sd = m.state_dict()
    assert "layer_model" in sd and any(k.startswith("base_model.") for k in sd)


def test_layer_batch_64_full_steps_properties():
    """Full size (batch 64, 400 steps): rows are independent (a batch equals its halves) and finite."""
    m = _model(LAYER_STEPS=400)
    gen = torch.Generator().manual_seed(9)
    E, start = torch.rand((64, 1), generator=gen).cuda(), torch.randn((64, 46), generator=gen).cuda()
    y = m.sample_layers(E, start=start, sample_offset=0)
    y2 = torch.cat([m.sample_layers(E[:32], start=start[:32], sample_offset=0),
                    m.sample_layers(E[32:], start=start[32:], sample_offset=0)])
    assert torch.isfinite(y).all() and torch.equal(y, y2)


def test_layer_model_other_widths_against_oracle():
    """HGCal-style layer model (3 conditioning inputs, 29 entries) and a wider / shallower one: forward, denoise for the three
    objectives' scalings and a DDim trajectory against the CPU oracle (no reference fixture: the oracle is pinned above)."""
    from calodiffusion_amd.resnet import ResNet
    from calodiffusion_amd import schedule
    for dim, cond_size, layers, hidden in ((29, 3, 5, 256), (46, 1, 3, 512), (7, 2, 2, 64)):
        torch.manual_seed(77)
        net = ResNet(dim_in=dim, num_layers=layers, hidden_dim=hidden, cond_size=cond_size)
        net._engine_opts = dict(time_kind="log", objective="hybrid", sigma_data=0.5)
        net.cuda()
        sd = {k: v.cpu() for k, v in net.state_dict().items()}
        gen = torch.Generator().manual_seed(3)
        B = 5
        x, c, tm = torch.randn((B, dim), generator=gen), torch.rand((B, cond_size), generator=gen), torch.randn((B,), generator=gen)
        with torch.no_grad():
            want = O.resnet_mlp_forward(sd, x, c, tm)
        assert rel_l2(net(x.cuda(), cond=c.cuda(), time=tm.cuda()).cpu().numpy(), want.numpy()) < TOL_OP, (dim, "forward")
        om = O.OracleLayerModel({"TIME_EMBED": "log", "TRAINING_OBJ": "hybrid_weight", "NOISE_SCHED": "linear"}, sd)
        sig = torch.tensor([50.0, 3.0, 0.7, 0.1, 0.01])
        with torch.no_grad():
            want = om.denoise(x, c, sig)
        got = net.engine().denoise(x.cuda(), sig.cuda(), c.cuda())
        assert rel_l2(got.cpu().numpy(), want.numpy()) < TOL_OP, (dim, "denoise")
        table = schedule.ddim_step_table(20, 0.0, 0)
        got, _, _ = net.engine().ddim_sample(x.cuda(), c.cuda(), table)
        with torch.no_grad():
            want, _, _ = om.ddim_sample(x, c, None, 20)
        assert rel_l2(got.cpu().numpy(), want.numpy()) < TOL_TRAJ, (dim, "ddim")


@pytest.mark.parametrize("lt", ["l2", "huber", "l1", "mse"])
def test_layer_model_training_step_against_autograd(lt):
    """LayerDiffusion.compute_loss in the layer state (layerdiffusion.py:52-57): loss and every parameter gradient of the layer
    model from one cd_layer_train_step_loss call against torch autograd through the oracle, for every LOSS_TYPE of Loss._loss
    (the reference's CI fixture trains with 'huber'); FusedAdam then steps the layer model."""
    from calodiffusion_amd.optim import FusedAdam
    m = _model(LOSS_TYPE=lt)
    assert m.loss_function.loss_type == lt
    gen = torch.Generator().manual_seed(21)
    B = 9
    layers = torch.randn((B, 46), generator=gen)
    E = torch.rand((B, 1), generator=gen)
    noise = torch.randn((B, 46), generator=gen)
    rnd = torch.randn((B,), generator=gen)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.layer_model.state_dict().items()}
    om = O.OracleLayerModel(m.config, sd)
    want = om.hybrid_l2_loss(layers, E, noise, rnd_normal=rnd, loss_type=lt)
    want.backward()
    m.set_layer_state(True)
    m.noise_generation = lambda shape: noise.cuda()
    params = dict(m.layer_model.named_parameters())
    opt = FusedAdam(m.layer_model.parameters(), lr=1e-3)
    opt.zero_grad()
    loss = m.compute_loss(None, E.cuda(), None, layers.cuda(), rnd_normal=rnd.cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(want.detach())) < 2e-6 * abs(float(want.detach()))
    num = den = 0.0
    worst = 0.0
    for k, p in params.items():
        g, w = p.grad.cpu().double(), sd[k].grad.double()
        num += float(((g - w) ** 2).sum()); den += float((w ** 2).sum())
        worst = max(worst, float(((g - w) ** 2).sum().sqrt() / (w ** 2).sum().sqrt().clamp_min(1e-30)))
    # (l1: the gradient is sign(d) / N -- an x0 within rounding of its target may flip one of the 414 signs)
    assert (num / den) ** 0.5 < (2e-3 if lt == "l1" else 5e-6) and worst < (2e-2 if lt == "l1" else 1e-4), (lt, (num / den) ** 0.5, worst)
    with torch.no_grad():
        assert abs(float(m.compute_loss(None, E.cuda(), None, layers.cuda(), rnd_normal=rnd.cuda())) - float(want.detach())) < 2e-6 * abs(float(want.detach()))
    before = m.layer_model.out_lay.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, m.layer_model.out_lay.weight)
    m.set_layer_state(False)
    assert all(p.grad is None for p in m.base_model.parameters())
