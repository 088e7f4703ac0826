"""GPU (MI355X): the training step (loss + every parameter gradient from cd_train_step) against torch autograd through the
CPU oracle, and one optimizer step through the reference's training-loop protocol."""
import numpy as np
import pytest
import torch

from conftest import gold, rel_l2
from helpers import t
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


def _model(name):
    from calodiffusion_amd.calodiffusion import CaloDiffusion
    from calodiffusion_amd.configs import load_config
    cfg = load_config(name)
    torch.manual_seed(1234)
    return CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"]), cfg


def _oracle_grads(cfg, sd, data, E, noise, layers, rnd, tsteps):
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    om = O.OracleModel(cfg, sd)
    loss = om.hybrid_l2_loss(data, E, noise, layers, rnd_normal=rnd, time=tsteps)
    loss.backward()
    return float(loss), {k: v.grad for k, v in om.sd.items()}


@pytest.mark.parametrize("name,B", [("tiny", 3), ("dataset2", 2), ("dataset3", 1)])
def test_parameter_gradients_match_autograd(name, B):
    m, cfg = _model(name)
    gen = torch.Generator().manual_seed(77)
    shape = [B] + list(cfg["SHAPE_PAD"][1:])
    data = torch.randn(shape, generator=gen)
    noise = torch.randn(shape, generator=gen)
    E = torch.rand((B, 3 if cfg.get("HGCAL") else 1), generator=gen)
    layers = torch.randn((B, 1 + cfg["SHAPE_FINAL"][2]), generator=gen) if "layer" in cfg["SHOWERMAP"] else None
    rnd = torch.randn((B,), generator=gen)
    tsteps = torch.randint(0, cfg["NSTEPS"], (B,), generator=gen)  # Dataset-3 (cosine schedule) draws sigma from the table
    sd_cpu = {k[6:]: v.detach().cpu() for k, v in m.state_dict().items()}
    want_loss, want = _oracle_grads(cfg, sd_cpu, data, E, noise, layers, rnd, tsteps)

    # Two identical steps: the plan's first training step reduces every weight gradient's partials where they arise and sizes the
    # region for them; from the second step on the reductions are queued and run as one launch (WgradReduceQueue): both against
    # the reference's gradients, and against each other.
    got_steps = []
    for rep in range(2):
        m.zero_grad()
        sigma = m.loss_function.draw_sigma(data.cuda(), time=tsteps.cuda(), rnd_normal=rnd.cuda())
        loss = m.loss_function.loss_function(m, data.cuda(), E.cuda(), sigma=sigma, noise=noise.cuda(),
                                             layers=None if layers is None else layers.cuda())
        assert loss.requires_grad and loss.dim() == 0
        assert abs(float(loss) - want_loss) <= 1e-5 * abs(want_loss)
        loss.backward()
        worst = []
        for kname, p in m.model.named_parameters():
            assert p.grad is not None, kname
            err = rel_l2(p.grad.cpu().numpy(), want[kname].numpy())
            worst.append((err, kname))
        worst.sort(reverse=True)
        print(f"[{name}] step {rep}: worst per-tensor gradient errors: {[(round(e, 8), k) for e, k in worst[:3]]}")
        assert worst[0][0] < 1e-4, worst[:8]
        # all gradients together
        got_all = np.concatenate([p.grad.cpu().numpy().ravel() for _, p in m.model.named_parameters()])
        want_all = np.concatenate([want[k].numpy().ravel() for k, _ in m.model.named_parameters()])
        print(f"[{name}] step {rep}: all gradients together: {rel_l2(got_all, want_all):.3e}")
        assert rel_l2(got_all, want_all) < 5e-6
        got_steps.append(got_all)
    assert rel_l2(got_steps[1], got_steps[0]) < 1e-6  # (the queued reduction sums short slot lists in another order)


def test_training_loop_protocol_one_adam_step():
    """zero_grad -> compute_loss -> backward -> Adam.step, as TrainDiffusion.training_loop does (train_diffusion.py:52-63);
    the updated weights are picked up by the next forward."""
    m, cfg = _model("tiny")
    opt = torch.optim.Adam(m.parameters(), lr=2e-5)  # small steps: the loss must go down monotonically on a fixed batch
    gen = torch.Generator().manual_seed(5)
    data = torch.randn((4, 1, 8, 8, 8), generator=gen).cuda()
    E, layers = torch.rand((4, 3), generator=gen).cuda(), torch.randn((4, 9), generator=gen).cuda()
    noise, rnd = torch.randn(data.shape, generator=gen).cuda(), torch.randn((4,), generator=gen).cuda()
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert np.isfinite(losses).all() and losses[3] < losses[2] < losses[1] < losses[0], losses
    with torch.no_grad():
        l_eval = m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)
    assert not l_eval.requires_grad and float(l_eval) < losses[0]


def test_fused_adam_matches_torch_adam():
    """cd_adam_step against torch.optim.Adam on the CPU (the optimizer of train/train.py:144): parameters, both moments and the
    checkpointable state after several steps, odd sizes, a learning-rate change in between, weight decay."""
    from calodiffusion_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(5)
    shapes = [(32, 32, 3, 3, 3), (96,), (7, 13), (1,), (64, 128)]
    for wd in (0.0, 1e-2):
        ref = [torch.randn(s, generator=gen).requires_grad_() for s in shapes]
        dev = [p.detach().clone().cuda().requires_grad_() for p in ref]
        o_ref = torch.optim.Adam(ref, lr=3e-3, weight_decay=wd)
        o_dev = FusedAdam(dev, lr=3e-3, weight_decay=wd)
        for it in range(6):
            for p, q in zip(ref, dev):
                g = torch.randn(p.shape, generator=gen) * (10.0 ** (it - 3))
                p.grad = g.clone()
                q.grad = g.clone().cuda()
            if it == 3:
                for o in (o_ref, o_dev):
                    o.param_groups[0]["lr"] = 1e-3
            o_ref.step()
            o_dev.step()
        for p, q in zip(ref, dev):
            assert rel_l2(q.detach().cpu().numpy(), p.detach().numpy()) < 1e-6
            assert rel_l2(o_dev.state[q]["exp_avg"].cpu().numpy(), o_ref.state[p]["exp_avg"].numpy()) < 1e-6
            assert rel_l2(o_dev.state[q]["exp_avg_sq"].cpu().numpy(), o_ref.state[p]["exp_avg_sq"].numpy()) < 1e-6
        # state_dict interchange with torch.optim.Adam
        o_t = torch.optim.Adam(dev, lr=1e-3, weight_decay=wd)
        o_t.load_state_dict(o_dev.state_dict())
        assert int(o_t.state[dev[0]]["step"]) == 6


def test_queued_weight_gradient_reductions_follow_the_batch_size():
    """The region that holds the weight gradients' partials until their single reduction launch is sized by the first training step
    of a plan and re-sized when a later step asks for more: batch 2, 2, 5, 5, 2 on one model -- the first step at each new, larger
    batch reduces (partly) where the partials arise, the following ones from the queue -- every step against autograd through
    the oracle."""
    m, cfg = _model("tiny")
    sd_cpu = {k[6:]: v.detach().cpu() for k, v in m.state_dict().items()}
    for step, B in enumerate((2, 2, 5, 5, 2)):
        gen = torch.Generator().manual_seed(100 + step)
        shape = [B] + list(cfg["SHAPE_PAD"][1:])
        data, noise = torch.randn(shape, generator=gen), torch.randn(shape, generator=gen)
        E = torch.rand((B, 3 if cfg.get("HGCAL") else 1), generator=gen)
        layers = torch.randn((B, 1 + cfg["SHAPE_FINAL"][2]), generator=gen) if "layer" in cfg["SHOWERMAP"] else None
        rnd = torch.randn((B,), generator=gen)
        tsteps = torch.randint(0, cfg["NSTEPS"], (B,), generator=gen)
        want_loss, want = _oracle_grads(cfg, sd_cpu, data, E, noise, layers, rnd, tsteps)
        m.zero_grad()
        sigma = m.loss_function.draw_sigma(data.cuda(), time=tsteps.cuda(), rnd_normal=rnd.cuda())
        loss = m.loss_function.loss_function(m, data.cuda(), E.cuda(), sigma=sigma, noise=noise.cuda(),
                                             layers=None if layers is None else layers.cuda())
        assert abs(float(loss) - want_loss) <= 1e-5 * abs(want_loss), (step, B)
        loss.backward()
        got_all = np.concatenate([p.grad.cpu().numpy().ravel() for _, p in m.model.named_parameters()])
        want_all = np.concatenate([want[k].numpy().ravel() for k, _ in m.model.named_parameters()])
        assert rel_l2(got_all, want_all) < 5e-6, (step, B, rel_l2(got_all, want_all))
