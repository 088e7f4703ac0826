"""GPU (MI355X): what round 2 added to the hot path, against fixtures made from the reference itself and the CPU oracle --
the other samplers as step programs (cd_sampler_run), sinusoidal embeddings, the fp16-range fallback of the sampler loops,
reference-generated gradients, Dataset-3 / HGCal trajectories and full-size properties, batch-sharded Philox streams, and the
training loop with FusedAdam."""
import copy

import numpy as np
import pytest
import torch

from conftest import gold, rel_l2
from helpers import seeded_unet, t, verify_checksums
from sampler_cases import CASES, options, replay_noise
from test_oracle_golden import check_sampler_case

pytestmark = pytest.mark.gpu

TOL_OP = 1e-5
TOL_TRAJ = 1e-4


def _model(name, over=None, seed=1234):
    from calodiffusion_amd.calodiffusion import CaloDiffusion
    from calodiffusion_amd.configs import load_config
    cfg = copy.deepcopy(load_config(name))
    cfg.update(over or {})
    torch.manual_seed(seed)
    return CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])


# ---------------------------------------------------------------------------------------------- samplers
@pytest.mark.parametrize("tag", sorted(CASES))
def test_sampler_programs_match_reference_trajectories(tag):
    """Every other sampler of models/sample.py as a step program on the device loop, against trajectories of the reference's
    own sampler classes (tiny config), the reference's noise draws replayed where the sampler is stochastic."""
    g = gold("samplers_tiny")
    name, over, _, off, rows = CASES[tag]
    cfg_over = dict(over, SAMPLER=name)
    opts = options(g, tag)
    if opts:
        cfg_over["SAMPLER_OPTIONS"] = opts
    m = _model("tiny", cfg_over)
    smp = m.sampler_algorithm
    assert type(smp).__name__ == name
    n = int(g[f"{tag}.n"])
    start, E, layers = (t(g[k])[:rows].cuda() for k in ("start", "E", "layers"))
    prog = smp.build(m, n, off).finalize()
    m.loss_function.update_step(m.nsteps)
    if name == "DPM":
        # DPM-Solver-fast cancels terms of order sigma_max (dpm_2: one second-order step from sigma = 142 to 1 turns a 4e-7 change
        # of a coefficient into 1.5e-4 of the end point), and torch's vectorised cos / exp / log / expm1 differ in the last bit
        # between CPUs: the reference itself, run on this host, ends 1.5e-4 from its own result in the container that made the
        # golden.  So this host's table must agree with the container's to rounding, and the run uses the container's
        # (tests/golden/dpm_tables.npz, oracle/gen_golden.py dpm_tables): every DPM case then holds north_star's 1e-4.
        gt = gold("dpm_tables")
        want = gt[f"{tag}.coefs"]
        assert prog.coefs.shape == want.shape and np.allclose(prog.coefs, want, rtol=2e-5, atol=0), tag
        assert abs(prog.start_scale - float(gt[f"{tag}.start_scale"])) <= 1e-6 * abs(prog.start_scale)
        fixed, scale = want.copy(), float(gt[f"{tag}.start_scale"])

        def build_with_container_table(model, num_steps, sample_offset, _b=smp.build):
            p = _b(model, num_steps, sample_offset)
            fin = p.finalize

            def finalize():
                fin()
                p.coefs, p.start_scale = fixed, scale
                return p
            p.finalize = finalize
            return p
        smp.build = build_with_container_table
    noise = replay_noise(g, tag, tuple(start.shape))
    if prog.n_randn:
        assert prog.n_randn == len(noise), (tag, prog.n_randn, len(noise))
        smp.step_noise = torch.stack(noise).cuda()
    out = m.sample(E, layers, num_steps=n, start=start, sample_offset=off, debug=True)
    x, xs, x0s = out
    to_np = lambda seq: None if seq is None or isinstance(seq, list) and not seq else [v.cpu().numpy() for v in seq]  # noqa: E731
    if name == "Consistency":
        x0s = None  # (the reference returns the last denoised tensor there, not a list)
    # every case holds north_star's 1e-4, dpm_2 included: DPM-Solver-fast runs as LINDIV ops in the reference's operation order on
    # the step table of the host that made the golden (above; round 3 had blamed the program's association and widened the bound
    # to 5e-4), so what is left on the device is the case's 9.8x amplification (tests/test_oracle_golden.py) of the denoise
    # kernels' own ~1e-6
    tol = TOL_TRAJ
    if tag == "dpm_2":
        print(f"dpm_2: device {rel_l2(np.asarray(x), g[f'{tag}.x']):.2e}")
    check_sampler_case(tag, g, x, to_np(xs), to_np(x0s), tol)


def test_sampler_program_graph_replay_equals_eager_and_philox():
    """A uniform program (Heun with churn) replays one captured step graph: bitwise equal to the eager run with the same
    Philox positions; a second trajectory re-uses the graph, draws new noise and differs."""
    outs = {}
    for use_graph in (True, False):
        m = _model("tiny", {"SAMPLER": "Heun", "NOISY_SAMPLE": True, "SAMPLER_OPTIONS": {"HIP_GRAPH": use_graph}})
        gen = torch.Generator().manual_seed(5)
        E, layers = torch.rand((4, 3), generator=gen).cuda(), torch.randn((4, 9), generator=gen).cuda()
        m.noise_offset = 0
        a = m.sample(E, layers, num_steps=6, debug=True)[1]  # xs: the last step of the reference's Heun is not finite
        b = m.sample(E, layers, num_steps=6, debug=True)[1]
        outs[use_graph] = (torch.stack(a).cpu(), torch.stack(b).cpu())
        assert torch.isfinite(outs[use_graph][0]).all()
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    assert not torch.equal(outs[True][0], outs[True][1])


def test_samplers_on_dataset2_and_by_name():
    from calodiffusion_amd import sample
    from calodiffusion_amd.utils import load_attr
    g = gold("samplers_dataset2")
    start, E, layers = t(g["start"]).cuda(), t(g["E"]).cuda(), t(g["layers"]).cuda()
    m = _model("dataset2", {"SAMPLER": "Heun"})
    _, xs, x0s = m.sample(E, layers, num_steps=4, start=start, debug=True)
    assert rel_l2(xs[3].cpu().numpy(), g["heun_xs3"]) < TOL_TRAJ and rel_l2(x0s[3].cpu().numpy(), g["heun_x0s3"]) < TOL_TRAJ
    m = _model("dataset2", {"SAMPLER": "LMS"})
    assert rel_l2(m.sample(E, layers, num_steps=6, start=start), g["lms_6"]) < TOL_TRAJ
    for name in ("DDim", "DDPM", "Euler", "Heun", "DPM2", "LMS", "Restart", "DPM", "DPMPP2S", "DPMPP2M", "Consistency", "DPMAdaptive",
                 "DPMPPSDE", "DPMPP2MSDE", "DPMPP3MSDE"):
        assert load_attr("sampler", name) is getattr(sample, name)
    for name in ("BespokeNonStationary",):
        with pytest.raises(NotImplementedError):
            load_attr("sampler", name)({})
    with pytest.raises(ValueError):
        load_attr("sampler", "NoSuchSampler")


# ---------------------------------------------------------------------------------------------- sinusoidal embeddings
def test_unet_sinusoidal_embeddings():
    """CondUnet(time_embed / cond_embed = True) (models.py:132-144, 578-601) through cd_unet_forward, against the reference's
    own CondUnet called directly; the denoise path refuses them like the reference's do_time_embed (KeyError for 'sin')."""
    g = gold("unet_sinusoidal")
    for tag, over in (("both", dict(time_embed=True, cond_embed=True, cond_size=1)),
                      ("time", dict(time_embed=True, cond_embed=False, cond_size=10)),
                      ("cond", dict(time_embed=False, cond_embed=True, cond_size=1))):
        kw = dict(out_dim=1, layer_sizes=[32, 32, 64, 32], channels=4, cond_dim=128, resnet_block_groups=8, mid_attn=True,
                  block_attn=True, compress_Z=True, cylindrical=True, data_shape=[1, 4, 8, 8, 8])
        kw.update(over)
        net = seeded_unet(kw, int(g["seed"])).cuda()
        y = net(t(g[f"{tag}.x"]).cuda(), cond=t(g[f"{tag}.cond"]).cuda(), time=t(g[f"{tag}.time"]).cuda())
        assert rel_l2(y.cpu().numpy(), g[f"{tag}.y"]) < TOL_OP, tag
    with pytest.raises(KeyError):
        _model("tiny", {"TIME_EMBED": "sin"})
    with pytest.raises(ValueError, match="cd_unet_forward only"):
        net.engine().denoise(torch.zeros(1, 1, 8, 8, 8, device="cuda"), torch.ones(1, device="cuda"), torch.zeros(1, 1, device="cuda"))


# ---------------------------------------------------------------------------------------------- fp16 range fallback
def test_sampler_recovers_from_fp16_range_overflow():
    """An activation beyond the fp16 range trips the f16x2 convolutions' flag; the sampler loops then re-run the trajectory
    with the exact bf16x3 convolutions instead of losing the batch: finite, and equal to a run in bf16x3 mode from the start."""
    m = _model("dataset2")
    gen = torch.Generator().manual_seed(8)
    start = torch.randn((2, 1, 45, 16, 9), generator=gen).cuda()
    E, layers = torch.rand((2, 1), generator=gen).cuda(), torch.randn((2, 46), generator=gen).cuda()
    with torch.no_grad():
        m.model.init_conv.conv.bias.fill_(1.0e6)  # drives the first block's conv input out of the fp16 range
    out = m.sample(E, layers, num_steps=3, start=start)
    assert np.isfinite(out).all() and m.engine().range_fallbacks == 1
    m.model.engine().check_status()  # the flag was consumed
    from calodiffusion_amd import sample
    m.sampler_algorithm = sample.Heun(dict(m.config, NOISY_SAMPLE=False))
    _, xs, _ = m.sample(E, layers, num_steps=3, start=start, debug=True)
    assert torch.isfinite(torch.stack(xs)).all() and m.engine().range_fallbacks == 2
    # the same trajectory with bf16x3 arithmetic from the start: identical, and no fallback
    from calodiffusion_amd import engine
    assert engine.get_conv_precision() == "f16x2"
    engine.set_conv_precision("bf16x3")
    try:
        m.sampler_algorithm = sample.DDim(m.config)
        want = m.sample(E, layers, num_steps=3, start=start)
        assert m.engine().range_fallbacks == 2
    finally:
        engine.set_conv_precision("f16x2")
    assert np.array_equal(out, want), float(np.abs(out - want).max())


# ---------------------------------------------------------------------------------------------- gradients
@pytest.mark.parametrize("name", ["dataset2", "dataset3", "hgcal"])
def test_gradients_match_the_reference(name):
    """cd_train_step against .grad of the reference's own compute_loss(...).backward() (fixtures: oracle/gen_golden.py grads /
    grads3; the HGCal fixture carries its own inputs)."""
    g = gold(f"grads_{name}")
    gl = g if name == "hgcal" else gold(f"loss_{name}")
    m = _model(name)
    data, E, noise = t(gl["data"]).cuda(), t(gl["E"]).cuda(), t(gl["noise"]).cuda()
    layers = t(gl["layers"]).cuda() if "layers" in gl.files else None
    if name in ("dataset2", "hgcal"):
        sigma = m.loss_function.draw_sigma(data, rnd_normal=t(gl["rnd_normal"]).cuda())
    else:
        sigma = m.loss_function.draw_sigma(data, time=torch.from_numpy(gl["time"]).cuda())
    m.zero_grad()
    loss = m.loss_function.loss_function(m, data, E, sigma=sigma, noise=noise, layers=layers)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    grads = dict(m.model.named_parameters())
    worst = 0.0
    for k in g.files:
        if k.startswith("grad."):
            err = rel_l2(grads[k[5:]].grad.cpu().numpy(), g[k])
            worst = max(worst, err)
            assert err < 1e-4, (name, k, err)
    for k, (s1, s2) in zip(g["ck_keys"], g["ck_vals"]):
        gr = grads[str(k)].grad.double()
        assert abs(float((gr * gr).sum()) - s2) <= 2e-4 * max(s2, 1e-30), (name, k)
    print(f"[{name}] worst whole-tensor gradient error vs the reference: {worst:.2e}")


def test_parameter_gradients_match_autograd_hgcal():
    """HGCal ([32,32,64,96], kZ = 4 up-convs, 4-channel init conv): every parameter gradient against autograd through the oracle."""
    from oracle import torch_oracle as O
    m = _model("hgcal")
    cfg = m.config
    gen = torch.Generator().manual_seed(79)
    shape = [1] + list(cfg["SHAPE_PAD"][1:])
    data, noise = torch.randn(shape, generator=gen), torch.randn(shape, generator=gen)
    E, layers = torch.rand((1, 3), generator=gen), torch.randn((1, 1 + cfg["SHAPE_FINAL"][2]), generator=gen)
    rnd = torch.randn((1,), generator=gen)
    sd = {k[6:]: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    om = O.OracleModel(cfg, sd)
    want_loss = om.hybrid_l2_loss(data, E, noise, layers, rnd_normal=rnd)
    want_loss.backward()
    m.zero_grad()
    loss = m.compute_loss(data.cuda(), E.cuda(), noise=noise.cuda(), layers=layers.cuda(), rnd_normal=rnd.cuda())
    loss.backward()
    assert abs(float(loss) - float(want_loss)) <= 1e-5 * abs(float(want_loss))
    errs = sorted(((rel_l2(p.grad.cpu().numpy(), om.sd[k].grad.numpy()), k) for k, p in m.model.named_parameters()), reverse=True)
    print("[hgcal] worst per-tensor gradient errors:", [(round(e, 8), k) for e, k in errs[:3]])
    assert errs[0][0] < 1e-4, errs[:6]


# ---------------------------------------------------------------------------------------------- trajectories, full sizes
def test_dataset3_ddim_and_hgcal_ddpm_trajectories():
    g = gold("ddim_dataset3")
    m = _model("dataset3")
    start, E = t(g["start"]).cuda(), t(g["E"]).cuda()
    for n in (10, 50):
        err = rel_l2(m.sample(E, None, num_steps=n, start=start), g[f"ddim_{n}"])
        assert err < TOL_TRAJ, (n, err)
    g = gold("ddpm_hgcal")
    m = _model("hgcal")
    assert type(m.sampler_algorithm).__name__ == "DDPM"
    start, E, layers = t(g["start"]).cuda(), t(g["E"]).cuda(), t(g["layers"]).cuda()
    torch.manual_seed(int(g["noise_seed"]))
    m.sampler_algorithm.step_noise = torch.stack([torch.randn(start.shape) for _ in range(200)]).cuda()
    out, xs, x0s = m.sample(E, layers, num_steps=200, start=start, debug=True)
    e100, e0, efin = rel_l2(xs[100].cpu().numpy(), g["x_step100"]), rel_l2(x0s[100].cpu().numpy(), g["x0_step100"]), rel_l2(out, g["ddpm_200"])
    print(f"[hgcal DDPM-200] x@100 {e100:.2e}  x0@100 {e0:.2e}  final {efin:.2e}")
    assert e100 < TOL_TRAJ and e0 < TOL_TRAJ and efin < TOL_TRAJ


@pytest.mark.parametrize("name,B", [("dataset3", 32), ("hgcal", 16)])
def test_full_size_properties(name, B):
    """BASELINE's batch sizes for Dataset-3 (32) and HGCal (16 per GPU): batch independence, determinism, and the oracle on
    two of the showers."""
    from oracle import torch_oracle as O
    m = _model(name)
    cfg = m.config
    gen = torch.Generator().manual_seed(13)
    x = torch.randn([B] + list(cfg["SHAPE_PAD"][1:]), generator=gen).cuda()
    E = torch.rand((B, 3 if cfg.get("HGCAL") else 1), generator=gen).cuda()
    layers = torch.randn((B, 1 + cfg["SHAPE_FINAL"][2]), generator=gen).cuda() if "layer" in cfg["SHOWERMAP"] else None
    sig = torch.full((B,), 0.8, device="cuda")
    y = m.denoise(x, E=E, sigma=sig, layers=layers)
    assert torch.isfinite(y).all()
    sl = slice(3, 8)
    y5 = m.denoise(x[sl].contiguous(), E=E[sl].contiguous(), sigma=sig[sl].contiguous(), layers=None if layers is None else layers[sl].contiguous())
    assert rel_l2(y5.cpu().numpy(), y[sl].cpu().numpy()) < 2e-6
    assert torch.equal(m.denoise(x, E=E, sigma=sig, layers=layers), y)
    om = O.OracleModel(cfg, {k: v.cpu() for k, v in m.state_dict().items()})
    pick = [0, B - 1]
    with torch.no_grad():
        want = om.denoise(x[pick].cpu(), E[pick].cpu(), sig[pick].cpu(), None if layers is None else layers[pick].cpu())
    assert rel_l2(y[pick].cpu().numpy(), want.numpy()) < TOL_OP


def test_ddpm_50_error_against_the_reference_is_recorded():
    """The stochastic tiny-config trajectory with the reference's seeded noise stream: observed error printed, held to the
    north_star's 1e-4."""
    g = gold("ddpm_tiny")
    m = _model("tiny")
    start, E, layers = t(g["start"]).cuda(), t(g["E"]).cuda(), t(g["layers"]).cuda()
    torch.manual_seed(int(g["noise_seed"]))
    m.sampler_algorithm.step_noise = torch.stack([torch.randn(start.shape) for _ in range(50)]).cuda()
    out = m.sample(E, layers, num_steps=50, start=start)
    err = rel_l2(out, g["ddpm_50"])
    print(f"[tiny DDPM-50] final rel L2 vs the reference: {err:.3e}")
    assert err < TOL_TRAJ


# ---------------------------------------------------------------------------------------------- sharded Philox streams
def test_batch_shards_draw_their_rows_of_one_global_stream():
    """with set_noise_shard every rank walks the same global Philox stream and draws only its rows, so the union of
    the shards is the single-GPU result of the same seed -- start noise bit for bit, and the stochastic (DDPM) trajectories."""
    from calodiffusion_amd.utils import shard_batch
    gen = torch.Generator().manual_seed(21)
    B = 6
    E, layers = torch.rand((B, 3), generator=gen).cuda(), torch.randn((B, 9), generator=gen).cuda()
    m = _model("tiny")
    m.noise_offset = 0
    full_start = m.noise_generation([B, 1, 8, 8, 8]).clone()
    m.noise_offset = 0
    full = [m.sample(E, layers, num_steps=7), m.sample(E, layers, num_steps=7)]
    off_full = m.noise_offset
    for world in (2, 3):
        parts = [[], []]
        for rank in range(world):
            sl = shard_batch(B, world, rank)
            ms = _model("tiny")
            ms.set_noise_shard(sl.start, B)
            ms.noise_offset = 0
            st = ms.noise_generation([sl.stop - sl.start, 1, 8, 8, 8])
            assert torch.equal(st, full_start[sl])
            ms.noise_offset = 0
            for k in range(2):
                parts[k].append(ms.sample(E[sl].contiguous(), layers[sl].contiguous(), num_steps=7))
            assert ms.noise_offset == off_full  # every rank ends at the same stream position
        for k in range(2):
            got = np.concatenate(parts[k])
            assert rel_l2(got, full[k]) < 2e-6, (world, k, rel_l2(got, full[k]))
    assert not np.allclose(full[0], full[1])


# ---------------------------------------------------------------------------------------------- training loop
def test_training_loop_with_fused_adam_updates_the_plan():
    """zero_grad -> compute_loss -> backward -> FusedAdam.step (train_diffusion.py:52-63 with the fused optimizer): the plan's
    packed weights follow the raw-pointer update (they did not), the loss falls, and the trajectory equals
    torch.optim.Adam's."""
    from calodiffusion_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(5)
    data = torch.randn((4, 1, 8, 8, 8), generator=gen).cuda()
    E, layers = torch.rand((4, 3), generator=gen).cuda(), torch.randn((4, 9), generator=gen).cuda()
    noise, rnd = torch.randn(data.shape, generator=gen).cuda(), torch.randn((4,), generator=gen).cuda()
    sig = torch.full((4,), 1.0, device="cuda")
    runs = {}
    for kind in ("fused", "torch"):
        m = _model("tiny")
        opt = (FusedAdam if kind == "fused" else torch.optim.Adam)(m.parameters(), lr=2e-5)
        y0 = m.denoise(data, E=E, sigma=sig, layers=layers).clone()
        losses = []
        for _ in range(4):
            opt.zero_grad()
            loss = m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        y1 = m.denoise(data, E=E, sigma=sig, layers=layers)
        assert losses[3] < losses[2] < losses[1] < losses[0], (kind, losses)
        assert not torch.equal(y0, y1)
        runs[kind] = (losses, y1.cpu().numpy(), [p.detach().cpu().numpy().copy() for p in m.parameters()])
    assert np.allclose(runs["fused"][0], runs["torch"][0], rtol=1e-5, atol=0), (runs["fused"][0], runs["torch"][0])
    assert rel_l2(runs["fused"][1], runs["torch"][1]) < 1e-5
    worst = max(rel_l2(a, b) for a, b in zip(runs["fused"][2], runs["torch"][2]))
    assert worst < 1e-5, worst


def test_generate_refuses_to_skip_the_inverse_preprocessing():
    m = _model("tiny")
    gen = torch.Generator().manual_seed(3)
    loader = [(torch.rand((2, 3), generator=gen), torch.randn((2, 9), generator=gen), None)]
    with pytest.raises(ValueError, match="inverse pre-processing"):
        m.generate(loader, sample_steps=2)
    raw, e = m.generate(loader, sample_steps=2, reverse_norm=False)
    assert raw.shape == (2, 1, 8, 8, 8)
    phys, _ = m.generate(loader, sample_steps=2, reverse_norm=lambda gen_, en, lay, cfg: (gen_ * 2.0, en))
    assert phys.shape == raw.shape


def test_all_weights_in_one_call_equals_one_tensor_at_a_time():
    """cd_plan_set_weights (every tensor by two launches from a job list) leaves the plan exactly as a cd_plan_set_weight per
    tensor does: the same denoise output, bit for bit, for the level shapes of Dataset-2 (all pack kinds: stride-1 / strided /
    transposed / 1x1 / init convs, raw tensors)."""
    import ctypes as C
    from calodiffusion_amd import engine as E_
    m = _model("dataset2")
    gen = torch.Generator().manual_seed(11)
    x = torch.randn((2, 1, 45, 16, 9), generator=gen).cuda()
    En, layers = torch.rand((2, 1), generator=gen).cuda(), torch.randn((2, 46), generator=gen).cuda()
    sig = torch.tensor([0.7, 3.0], device="cuda")
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.0 + 0.01 * torch.randn(p.shape, generator=gen).to(p.device))  # (changes every tensor: bumps the versions)
    y_batch = m.denoise(x, E=En, sigma=sig, layers=layers).clone()  # sync_weights -> cd_plan_set_weights
    eng = m.engine()
    sd = m.model.state_dict() if hasattr(m, "model") else eng.unet.state_dict()
    for name, tns in eng._weight_order:
        tt = tns.detach().float().contiguous()
        E_._check(eng.lib.cd_plan_set_weight(eng.plan, name.encode(), tt.data_ptr(), tt.numel(), E_._stream()))
    assert len(sd) == len(eng._weight_order)
    y_one = m.denoise(x, E=En, sigma=sig, layers=layers)
    assert torch.equal(y_batch, y_one)
