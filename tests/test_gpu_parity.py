"""GPU (MI355X): the HIP path, through the C ABI, against the golden vectors of the reference and the CPU oracle.

Tolerances: north_star asks for <= 1e-4 relative L2 of the reference CPU path.  Single kernels / single denoise calls are
held to 1e-5 (the fp32 reorder floor of the reference itself is 5.5e-7 per denoise call); multi-step
trajectories to 1e-4.
"""
import os

import numpy as np
import pytest
import torch

from conftest import gold, rel_l2
from helpers import d1grid_kwargs, seeded_unet, t, verify_checksums

pytestmark = pytest.mark.gpu

TOL_OP = 1e-5
TOL_TRAJ = 1e-4


@pytest.fixture(scope="module")
def ops():
    from calodiffusion_amd.engine import Ops
    return Ops()


def cl(ops, a):
    return ops.to_channels_last(t(a).cuda())


def back(ops, y_cl):
    return ops.to_ncdhw(y_cl).cpu().numpy()


def test_library_is_the_hip_one_and_device_is_gfx950():
    from calodiffusion_amd import engine
    name = engine.require_gpu()
    assert "gfx950" in name


def test_layout_roundtrip(ops):
    x = torch.randn(3, 32, 4, 5, 6)
    y = ops.to_channels_last(x.cuda())
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 4, 1).contiguous())
    assert torch.equal(ops.to_ncdhw(y).cpu(), x)


def test_cyl_conv_known_answer(ops):
    """calodiffusion/tests/test_cyl_conv.py embedded in channel 0 of a 32-channel problem (1x3x3 kernel -> zero d-taps)."""
    g = gold("cyl_known_answer")
    x = torch.zeros(1, 32, 1, 4, 3)
    x[:, 0] = t(g["x"])[:, 0]
    w = torch.zeros(32, 32, 3, 3, 3)
    w[0, 0, 1] = 1.0  # all-ones 1x3x3 kernel in the central z plane
    y = back(ops, ops.cyl_conv(cl(ops, x.numpy()), w.cuda(), None))
    assert np.array_equal(y[0, 0, 0], g["cyl"][0, 0, 0])
    assert np.count_nonzero(y[0, 1:]) == 0


def test_conv_primitives(ops):
    g = gold("primitives_conv")
    tags = sorted({k.split(".")[0] for k in g.files})
    for tag in tags:
        w = t(g[f"{tag}.w"]).cuda()
        b = t(g[f"{tag}.b"]).cuda() if f"{tag}.b" in g.files else None
        if tag in ("c3_4_32", "c3_3_32"):
            y = back(ops, ops.init_conv(t(g[f"{tag}.x"]).cuda(), w, b))
        elif tag.startswith("c3") or tag.startswith("c1"):
            if tag == "c1_32_1":
                continue  # single-output head is covered by the fused head kernel in the model tests
            y = back(ops, ops.cyl_conv(cl(ops, g[f"{tag}.x"]), w, b))
        elif tag.startswith("down"):
            zs = 2 if int(g[f"{tag}.cz"]) else 1
            y = back(ops, ops.cyl_conv(cl(ops, g[f"{tag}.x"]), w, b, stride=(zs, 2, 2)))
        else:
            zs = 2 if int(g[f"{tag}.cz"]) else 1
            e = g[f"{tag}.extra"]
            y = back(ops, ops.cyl_conv_transpose(cl(ops, g[f"{tag}.x"]), w, b, int(w.shape[2]), zs, (0, int(e[1]), int(e[2]))))
        assert y.shape == g[f"{tag}.y"].shape, (tag, y.shape, g[f"{tag}.y"].shape)
        err = rel_l2(y, g[f"{tag}.y"])
        assert err < TOL_OP, (tag, err)


def test_halo_tiled_kernels_match_too(ops, monkeypatch):
    """Grids whose planes do not fit the whole-plane LDS image (Dataset-3 level 0) take the (TZ, TH) halo-tiled kernels;
    force them on the small golden cases."""
    monkeypatch.setenv("CD_NO_FLAT", "1")
    g = gold("primitives_conv")
    for tag in ("c3_32_32", "c3_64_32", "c3_96_64", "down_d2", "down_odd", "down_noz"):
        w = t(g[f"{tag}.w"]).cuda()
        b = t(g[f"{tag}.b"]).cuda()
        stride = (1, 1, 1)
        if tag.startswith("down"):
            stride = (2 if int(g[f"{tag}.cz"]) else 1, 2, 2)
        y = back(ops, ops.cyl_conv(cl(ops, g[f"{tag}.x"]), w, b, stride=stride))
        err = rel_l2(y, g[f"{tag}.y"])
        assert err < TOL_OP, (tag, err)


def test_concat_conv_equals_conv_of_concat(ops):
    """The skip concat is never materialised: the conv reads two base pointers (models.py:741)."""
    from oracle import torch_oracle as O
    gen = torch.Generator().manual_seed(3)
    a, b = torch.randn(2, 64, 5, 4, 3, generator=gen), torch.randn(2, 64, 5, 4, 3, generator=gen)
    w, bias = torch.randn(32, 128, 3, 3, 3, generator=gen) * 0.05, torch.randn(32, generator=gen)
    want = O.cyl_conv3d(torch.cat([a, b], 1), w, bias, padding=(1, 1, 1)).numpy()
    y = back(ops, ops.cyl_conv(cl(ops, a.numpy()), w.cuda(), bias.cuda(), x1_cl=cl(ops, b.numpy())))
    assert rel_l2(y, want) < TOL_OP
    w1 = torch.randn(32, 128, 1, 1, 1, generator=gen) * 0.1
    want = O.cyl_conv3d(torch.cat([a, b], 1), w1, bias).numpy()
    y = back(ops, ops.cyl_conv(cl(ops, a.numpy()), w1.cuda(), bias.cuda(), x1_cl=cl(ops, b.numpy())))
    assert rel_l2(y, want) < TOL_OP


def _conv_case(ops, gen, B, c0, c1, cout, shape, nb=3):
    from oracle import torch_oracle as O
    cin = c0 + c1
    x = torch.randn((B, cin) + shape, generator=gen)
    w, bias = torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.05, torch.randn(cout, generator=gen)
    nb = B if nb is None else min(B, nb)
    want = O.cyl_conv3d(x[:nb], w, bias, padding=(1, 1, 1)).numpy()
    if c1:
        y = ops.cyl_conv(cl(ops, x[:, :c0].contiguous().numpy()), w.cuda(), bias.cuda(), x1_cl=cl(ops, x[:, c0:].contiguous().numpy()))
    else:
        y = ops.cyl_conv(cl(ops, x.numpy()), w.cuda(), bias.cuda())
    return rel_l2(back(ops, y)[:nb], want)


def test_full_resolution_conv_kernel(ops):
    """The z-slide f16x2 kernel (kernels_conv_zs.hip) takes 3x3x3 convs on grids whose planes hold 128..160 voxels: Dataset-2's
    level 0 (16x9) in every channel configuration the U-Net uses there, ragged chunk ends, one and many samples."""
    gen = torch.Generator().manual_seed(11)
    for B, c0, c1, cout, shape in ((1, 32, 0, 32, (45, 16, 9)), (3, 32, 0, 32, (45, 16, 9)), (2, 32, 32, 32, (10, 16, 9)),
                                   (2, 32, 0, 64, (7, 16, 8)), (5, 64, 0, 32, (3, 16, 9)), (64, 32, 0, 32, (45, 16, 9))):
        err = _conv_case(ops, gen, B, c0, c1, cout, shape, nb=None)  # every sample, the 64 of the headline shape included
        assert err < 2e-6, (B, c0, c1, cout, shape, err)
    # wider planes run as phi strips with halo rows and a 5-plane ring: Dataset-3 (50x18 -> 10 strips of 5 rows), HGCal
    # (12x21 -> 3 strips of 4 rows), a two-strip grid with a concatenated input
    for B, c0, c1, cout, shape in ((1, 32, 0, 32, (6, 50, 18)), (2, 32, 0, 32, (9, 12, 21)), (1, 32, 0, 32, (45, 50, 18)),
                                   (2, 32, 32, 32, (5, 10, 18))):
        err = _conv_case(ops, gen, B, c0, c1, cout, shape)
        assert err < 2e-6, (B, c0, c1, cout, shape, err)


def test_wide_plane_images_of_the_full_resolution_conv(ops):
    """Round 3: plane images of up to 224 and 256 voxels (7 / 8 staging pieces per thread).  Dataset-3's level 1 (25x9 = 225-voxel
    planes, whole), whole planes of 161..224 voxels, HGCal's strips of 6 + 2 rows of 21, a strip grid of 10 + 2 rows with a
    concatenated input and a 64-channel output, ragged depths; every sample compared."""
    gen = torch.Generator().manual_seed(13)
    for B, c0, c1, cout, shape in ((3, 32, 0, 32, (23, 25, 9)), (2, 32, 32, 32, (7, 25, 9)), (2, 32, 0, 64, (5, 14, 13)),
                                   (2, 32, 0, 32, (11, 12, 17)), (3, 32, 0, 32, (28, 12, 21)), (2, 32, 32, 64, (6, 20, 18)),
                                   (2, 64, 0, 32, (4, 16, 16)), (1, 32, 0, 32, (3, 8, 32))):
        err = _conv_case(ops, gen, B, c0, c1, cout, shape, nb=None)
        assert err < 2e-6, (B, c0, c1, cout, shape, err)


def test_strip_conv_many_samples_every_sample_checked(ops):
    """Regression: with many workgroups in flight a staged plane was once converted before its loads had landed (the count
    of vector-memory operations the kernel's `s_waitcnt vmcnt(n)` relies on had been changed by the compiler): wrong rows in
    a few random samples of a large batch only.  Every sample of a 16-shower Dataset-3 batch is compared."""
    gen = torch.Generator().manual_seed(12)
    for B, c0, c1, cout, shape in ((16, 32, 0, 32, (45, 50, 18)), (24, 32, 32, 32, (12, 16, 9))):
        err = _conv_case(ops, gen, B, c0, c1, cout, shape, nb=B)
        assert err < 2e-6, (B, c0, c1, cout, shape, err)


def test_batched_denoise_equals_single_samples_dataset3():
    """Showers are independent: a Dataset-3 batch of 12 equals its showers denoised one at a time (catches races that only
    show with many workgroups in flight)."""
    m = _model("dataset3")
    gen = torch.Generator().manual_seed(5)
    B = 12
    x = torch.randn([B] + list(m.config["SHAPE_PAD"][1:]), generator=gen).cuda()
    E = torch.rand((B, 1), generator=gen).cuda()
    sig = torch.full((B,), 0.9, device="cuda")
    y = m.denoise(x, E=E, sigma=sig, layers=None)
    assert torch.isfinite(y).all()
    for i in (0, 5, 11):
        yi = m.denoise(x[i:i + 1], E=E[i:i + 1], sigma=sig[i:i + 1], layers=None)
        assert rel_l2(y[i:i + 1].cpu().numpy(), yi.cpu().numpy()) < 2e-6, i


def test_whole_sample_conv_kernel(ops):
    """Grids of at most 128 voxels per sample (Dataset-2's level 2: 12x4x2) take the whole-sample-in-LDS kernel
    (kernels_conv_small.hip): single-row / single-column grids, >64 input channels (two staging passes), concatenated inputs."""
    gen = torch.Generator().manual_seed(12)
    for B, c0, c1, cout, shape in ((2, 32, 0, 32, (12, 4, 2)), (3, 64, 0, 64, (12, 4, 2)), (2, 32, 32, 32, (12, 4, 2)),
                                   (1, 128, 0, 64, (5, 5, 5)), (2, 96, 0, 32, (3, 1, 4)), (2, 32, 0, 32, (1, 2, 3)),
                                   (2, 64, 64, 96, (4, 4, 8))):
        err = _conv_case(ops, gen, B, c0, c1, cout, shape)
        assert err < 2e-6, (B, c0, c1, cout, shape, err)


def test_conv_precision_modes_agree(ops, monkeypatch):
    """CD_CONV_PRECISION selects the arithmetic of the MFMA convs: f16x2 (default), bf16x3, f32.  All are fp32-grade."""
    gen = torch.Generator().manual_seed(13)
    errs = {}
    for mode in ("bf16x3", "f32"):
        monkeypatch.setenv("CD_CONV_PRECISION", mode)
        # the mode is latched at the first conv launch of a process: run the comparison in a child interpreter
        import subprocess, sys, json
        code = ("import sys, json, torch; sys.path.insert(0, %r); sys.path.insert(0, %r);"
                "from test_gpu_parity import _conv_case; from calodiffusion_amd.engine import Ops;"
                "g = torch.Generator().manual_seed(13);"
                "print(json.dumps([_conv_case(Ops(), g, 2, 32, 0, 32, (9, 16, 9)), _conv_case(Ops(), g, 2, 64, 0, 64, (6, 4, 2))]))"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        errs[mode] = json.loads(out.stdout.strip().splitlines()[-1])
        assert max(errs[mode]) < 3e-6, (mode, errs[mode])


def test_fp16_range_flag_and_fallback_of_plain_denoise():
    """The default arithmetic (f16x2 convs, fp16-pipe attention products) needs |operand| <= 65504.  Leaving that range
    poisons the raw cd_denoise output with inf/NaN AND raises the plan's sticky flag; denoise() as the samplers' callback
    (cd_denoise_safe) recovers instead: the call is re-run with the full-range kernels and equals the bf16x3 result."""
    from calodiffusion_amd import engine
    m = _model("dataset2")
    cfg = m.config
    B = 2
    shape = [B] + list(cfg["SHAPE_PAD"][1:])
    x = torch.randn(shape, device="cuda")
    E = torch.rand((B, 1), device="cuda")
    layers = torch.randn((B, cfg["SHAPE_PAD"][2] + 1), device="cuda")
    sig = torch.full((B,), 1.0, device="cuda")
    eng = m.engine()
    eng.safe_denoise = False  # the raw, asynchronous entry point
    m.denoise(x, E=E, sigma=sig, layers=layers)
    eng.check_status()  # in range: no exception
    with torch.no_grad():
        m.model.init_conv.conv.bias.fill_(1.0e6)  # drives the first block's conv input out of range
    bad = m.denoise(x, E=E, sigma=sig, layers=layers)
    assert not torch.isfinite(bad).all()
    with pytest.raises(FloatingPointError):
        eng.check_status()
    eng.check_status()  # the flag was cleared by the query
    # a flag left behind by an un-queried raw call is neither consumed by the next safe call nor mistaken for its own
    m.denoise(x, E=E, sigma=sig, layers=layers)
    eng.safe_denoise = True
    before = getattr(eng, "range_fallbacks", 0)
    got = m.denoise(x, E=E, sigma=sig, layers=layers)
    assert torch.isfinite(got).all() and eng.range_fallbacks == before + 1
    with pytest.raises(FloatingPointError):
        eng.check_status()  # bit 0 of the EARLIER raw call is still reported
    engine.set_conv_precision("bf16x3")
    try:
        want = m.denoise(x, E=E, sigma=sig, layers=layers)
        assert eng.range_fallbacks == before + 1  # full-range arithmetic up front: nothing to fall back from
    finally:
        engine.set_conv_precision("f16x2")
    assert torch.equal(got, want)
    # in range again: the safe call takes no fallback
    with torch.no_grad():
        m.model.init_conv.conv.bias.fill_(0.0)
    ok = m.denoise(x, E=E, sigma=sig, layers=layers)
    assert torch.isfinite(ok).all() and eng.range_fallbacks == before + 1
    eng.check_status()


def test_trained_scale_residual_stream_needs_no_fallback():
    """Random-init weights keep the un-normalised residual stream at O(1); trained checkpoints do not.  The level-0 convs'
    biases and weights are scaled so that the stream reaches ~1e3-1e4 (GroupNorm renormalises inside the blocks; the stream
    itself -- shortcut sums, attention inputs, the f16x2-staged conv inputs -- carries the magnitude): denoise must still
    equal the fp32 oracle to 1e-5 without leaving the fp16 range."""
    from oracle import torch_oracle as O
    m = _model("dataset2")
    with torch.no_grad():
        sd = m.model.state_dict()
        sd["init_conv.conv.weight"].mul_(3.0e3)
        sd["init_conv.conv.bias"].mul_(3.0e3)
        for k in sd:  # the blocks' own outputs follow the stream's scale through their closing norms' affine parameters
            if k.endswith("block2.norm.weight") or k.endswith("block2.norm.bias") or k.endswith("to_out.1.weight"):
                sd[k].mul_(1.0e3)
    g = torch.Generator().manual_seed(21)
    B = 2
    x = torch.randn([B] + list(m.config["SHAPE_PAD"][1:]), generator=g)
    E, layers = torch.rand((B, 1), generator=g), torch.randn((B, m.config["SHAPE_PAD"][2] + 1), generator=g)
    sig = torch.tensor([1.0, 30.0])
    om = O.OracleModel(m.config, {k: v.detach().cpu() for k, v in m.state_dict().items()})
    seen, orig = [], O.resnet_block
    try:  # max |input| of every ResnetBlock of the oracle's forward = the residual stream
        O.resnet_block = lambda sd_, p, xx, *a: (seen.append(float(xx.abs().max())), orig(sd_, p, xx, *a))[1]
        with torch.no_grad():
            want = om.denoise(x, E, sig, layers)
    finally:
        O.resnet_block = orig
    stream = max(seen)
    assert len(seen) == 15 and min(seen) > 1e3 and stream < 6e4, seen
    eng = m.engine()
    before = getattr(eng, "range_fallbacks", 0)
    got = m.denoise(x.cuda(), E=E.cuda(), sigma=sig.cuda(), layers=layers.cuda()).cpu()
    assert getattr(eng, "range_fallbacks", 0) == before
    err = float((got - want).norm() / want.norm())
    print(f"trained-scale stream (block inputs {min(seen):.0f} .. {stream:.0f}): denoise rel L2 {err:.2e}")
    assert err < 1e-5


def test_group_norm(ops):
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(4)
    for C, G, shp in ((32, 8, (2, 5, 6, 4)), (64, 8, (1, 23, 8, 4)), (96, 8, (1, 3, 3, 5)), (64, 1, (2, 7, 3, 5))):
        x = torch.randn((shp[0], C) + shp[1:], generator=gen) * 3 + 1.5
        gm, bt = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
        add, res = torch.randn(shp[0], C, generator=gen), torch.randn(x.shape, generator=gen)
        want = F.silu(F.group_norm(x, G, gm, bt, 1e-5)) + add[:, :, None, None, None] + res
        y = back(ops, ops.group_norm(cl(ops, x.numpy()), gm.cuda(), bt.cuda(), G, silu=True, add_bc=add.cuda(),
                                     residual=cl(ops, res.numpy())))
        assert rel_l2(y, want.numpy()) < TOL_OP, (C, G)


def _sub(g, tag):
    pre = f"{tag}.sd."
    return {k[len(pre):]: t(g[k]).cuda() for k in g.files if k.startswith(pre)}


def test_resnet_blocks(ops):
    g = gold("primitives_blocks")
    for tag in ("res_32_32", "res_32_64", "res_128_32", "res_nocond"):
        cond = t(g[f"{tag}.cond"]).cuda() if f"{tag}.cond" in g.files else None
        x = g[f"{tag}.x"]
        if tag == "res_128_32":  # exercise the two-pointer concat path
            y = ops.resnet_block(cl(ops, x[:, :64]), _sub(g, tag), cond, x1_cl=cl(ops, x[:, 64:]))
        else:
            y = ops.resnet_block(cl(ops, x), _sub(g, tag), cond)
        err = rel_l2(back(ops, y), g[f"{tag}.y"])
        assert err < TOL_OP, (tag, err)


def test_linear_attention_blocks(ops):
    g = gold("primitives_blocks")
    for tag in ("attn_32", "attn_64", "attn_96"):
        y = ops.linear_attention(cl(ops, g[f"{tag}.x"]), _sub(g, tag))
        err = rel_l2(back(ops, y), g[f"{tag}.y"])
        assert err < TOL_OP, (tag, err)


def _model(name):
    from calodiffusion_amd.calodiffusion import CaloDiffusion
    from calodiffusion_amd.configs import load_config
    cfg = load_config(name)
    torch.manual_seed(1234)
    m = CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
    return m


@pytest.mark.parametrize("name", ["tiny", "dataset2", "hgcal", "dataset3"])
def test_denoise_matches_reference(name):
    g = gold(f"model_{name}")
    m = _model(name)
    verify_checksums({k[6:]: v.cpu() for k, v in m.state_dict().items()}, g)
    x, E = t(g["x"]).cuda(), t(g["E"]).cuda()
    layers = t(g["layers"]).cuda() if "layers" in g.files else None
    for i in range(3):
        s = float(g[f"sigma_{i}"])
        sig = torch.full((x.shape[0], 1, 1, 1, 1), s, device="cuda")
        y = m.denoise(x * float(np.sqrt(1.0 + s * s)), E=E, sigma=sig, layers=layers)
        err = rel_l2(y.cpu().numpy(), g[f"denoise_{i}"])
        assert err < TOL_OP, (name, i, err)


def test_unet_forward_d1_grid():
    g = gold("unet_d1grid")
    net = seeded_unet(d1grid_kwargs(), int(g["seed"])).cuda()
    y = net(t(g["x"]).cuda(), cond=t(g["cond"]).cuda(), time=t(g["time"]).cuda())
    assert rel_l2(y.cpu().numpy(), g["y"]) < TOL_OP


def test_ddim_trajectories_dataset2():
    g = gold("ddim_dataset2")
    m = _model("dataset2")
    start, E, layers = t(g["start"]).cuda(), t(g["E"]).cuda(), t(g["layers"]).cuda()
    for n in (2, 10, 50, 400):
        out = m.sample(E, layers, num_steps=n, start=start)
        assert isinstance(out, np.ndarray) and out.dtype == np.float32 and out.shape == tuple(start.shape)
        err = rel_l2(out, g[f"ddim_{n}"])
        assert err < TOL_TRAJ, (n, err)
    out, xs, x0s = m.sample(E, layers, num_steps=10, start=start, debug=True)
    assert rel_l2(torch.stack(xs).cpu().numpy(), g["ddim_10_xs"]) < TOL_TRAJ
    assert rel_l2(torch.stack(x0s).cpu().numpy(), g["ddim_10_x0s"]) < TOL_TRAJ
    assert rel_l2(m.sample(E, layers, num_steps=10, start=start, sample_offset=3), g["ddim_10_off3"]) < TOL_TRAJ


def test_reverse_norm_on_device():
    """cd_reverse_norm (inverse logit-norm, per-layer rescale to the conditioning layer energies, energy scale, threshold)
    against the reference's ReverseNormCaloChall; also through the reference-shaped entry point ReverseNorm(...)."""
    from calodiffusion_amd.postprocess import ReverseNorm
    g = gold("reverse_norm")
    for tag, dnum, smap in (("d2", 2, "layer-logit-norm"), ("d3", 3, "logit-norm")):
        layerE = g[f"{tag}.layerE"] if f"{tag}.layerE" in g.files else None
        data, energy = ReverseNorm(g[f"{tag}.vox"], g[f"{tag}.e"], emax=1000., emin=1., max_deposit=2, logE=True, layerE=layerE,
                                   showerMap=smap, dataset_num=dnum, ecut=0.0000151)
        assert data.shape == g[f"{tag}.data"].shape and data.dtype == np.float32
        assert np.array_equal(np.asarray(energy, dtype=np.float32), g[f"{tag}.energy"])
        assert rel_l2(data, g[f"{tag}.data"]) < 1e-5, tag
        assert ((data == 0) == (g[f"{tag}.data"] == 0)).mean() > 0.999
    with pytest.raises(NotImplementedError):
        ReverseNorm(g["d2.vox"], g["d2.e"], showerMap="log-norm", dataset_num=2)


def test_generate_returns_physical_showers():
    """Diffusion.generate (diffusion.py:118-197): sampling loop over a loader + inverse pre-processing on the device; the
    output has the reference's shapes (SHAPE_ORIG, (N, 1)) and equals ReverseNorm applied to the normalised-space showers."""
    from calodiffusion_amd.postprocess import ReverseNorm
    m = _model("dataset2")
    cfg = m.config
    gen = torch.Generator().manual_seed(3)
    loader = [(torch.rand((3, 1), generator=gen), torch.randn((3, 46), generator=gen), None) for _ in range(2)]
    m.noise_offset = 0
    raw, e_raw = m.generate(loader, sample_steps=4, reverse_norm=False)
    m.noise_offset = 0
    phys, e = m.generate(loader, sample_steps=4)
    assert raw.shape == (6, 1, 45, 16, 9) and phys.shape == (6, 6480) and e.shape == (6, 1)
    layers = np.concatenate([l.numpy() for _, l, _ in loader])
    want, want_e = ReverseNorm(raw, e_raw, emax=cfg["EMAX"], emin=cfg["EMIN"], layerE=layers, logE=cfg["logE"],
                               max_deposit=cfg["MAXDEP"], showerMap=cfg["SHOWERMAP"], dataset_num=2, ecut=float(cfg["ECUT"]))
    assert np.array_equal(phys, want) and np.array_equal(e, np.reshape(want_e, (6, -1)))
    assert np.isfinite(phys).all() and (phys >= 0).all()


def test_edm_euler_sampler_dataset2():
    """The EDM Euler sampler rides the device sampler loop (one step table, one captured step graph); against the
    reference's own Euler trajectories."""
    from calodiffusion_amd import sample
    g = gold("euler_dataset2")
    m = _model("dataset2")
    start, E, layers = t(g["start"]).cuda(), t(g["E"]).cuda(), t(g["layers"]).cuda()
    eul = sample.Euler(m.config)
    for n in (5, 18):
        x, _, _ = eul(m, start, E, layers, n, 0, False)
        assert rel_l2(x.cpu().numpy(), g[f"euler_{n}"]) < TOL_TRAJ, n
    x, xs, x0s = eul(m, start, E, layers, 5, 0, True)
    assert rel_l2(torch.stack(x0s).cpu().numpy(), g["euler_5_x0s"]) < TOL_TRAJ
    x, _, _ = eul(m, start, E, layers, 18, 2, False)
    assert rel_l2(x.cpu().numpy(), g["euler_18_off2"]) < TOL_TRAJ
    # resolved by name like the reference's utils.load_attr("sampler", ...)
    from calodiffusion_amd.utils import load_attr
    assert load_attr("sampler", "Euler") is sample.Euler


def test_graph_replay_equals_eager_bitwise():
    m = _model("tiny")
    g = gold("ddpm_tiny")
    start, E, layers = t(g["start"]).cuda(), t(g["E"]).cuda(), t(g["layers"]).cuda()
    from calodiffusion_amd import sample
    m.sampler_algorithm = sample.DDim(m.config)
    m.sampler_algorithm.use_graph = True
    a = m.sample(E, layers, num_steps=20, start=start)
    b = m.sample(E, layers, num_steps=20, start=start)  # cached graph
    m.sampler_algorithm.use_graph = False
    c = m.sample(E, layers, num_steps=20, start=start)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert np.isfinite(a).all()


def test_stochastic_sampler_graph_replay_equals_eager_bitwise():
    """DDPM draws its per-step noise from the device Philox stream at a position read from the step counter, so one captured
    step graph serves a stochastic trajectory too: same (seed, offset) => graph and eager runs are bitwise identical, and a
    second trajectory (new offset) re-uses the graph and differs."""
    from calodiffusion_amd import sample
    m = _model("tiny")
    g = gold("ddpm_tiny")
    start, E, layers = t(g["start"]).cuda(), t(g["E"]).cuda(), t(g["layers"]).cuda()
    outs = {}
    for use_graph in (True, False):
        cfg = dict(m.config)
        cfg["SAMPLER_OPTIONS"] = {"HIP_GRAPH": use_graph, "SEED": 7}
        ddpm = sample.DDPM(cfg)
        m.noise_offset = 1000
        x1, _, _ = ddpm(m, start, E, layers, 20, 0, False)
        m.noise_offset = 999000
        x2, _, _ = ddpm(m, start, E, layers, 20, 0, False)
        outs[use_graph] = (x1.clone(), x2.clone())
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    assert not torch.equal(outs[True][0], outs[True][1]) and torch.isfinite(outs[True][1]).all()


def test_ddpm_tiny_with_reference_noise_stream():
    g = gold("ddpm_tiny")
    m = _model("tiny")
    assert type(m.sampler_algorithm).__name__ == "DDPM"
    start, E, layers = t(g["start"]).cuda(), t(g["E"]).cuda(), t(g["layers"]).cuda()
    torch.manual_seed(int(g["noise_seed"]))
    noise = torch.stack([torch.randn(start.shape) for _ in range(50)]).cuda()
    m.sampler_algorithm.step_noise = noise
    out, xs, x0s = m.sample(E, layers, num_steps=50, start=start, debug=True)
    assert rel_l2(xs[10].cpu().numpy(), g["x_step10"]) < TOL_TRAJ
    assert rel_l2(x0s[10].cpu().numpy(), g["x0_step10"]) < TOL_TRAJ
    assert rel_l2(out, g["ddpm_50"]) < TOL_TRAJ  # 50 stochastic steps, the reference's noise stream: north_star's 1e-4
    # device Philox noise: runs, finite, and differs from the fixed-noise result
    m.sampler_algorithm.step_noise = None
    out2 = m.sample(E, layers, num_steps=50, start=start)
    assert np.isfinite(out2).all() and not np.allclose(out2, out)


def test_loss_values():
    g = gold("loss_dataset2")
    m = _model("dataset2")
    loss = m.compute_loss(t(g["data"]).cuda(), t(g["E"]).cuda(), noise=t(g["noise"]).cuda(), layers=t(g["layers"]).cuda(),
                          rnd_normal=t(g["rnd_normal"]).cuda())
    assert loss.dim() == 0
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    g = gold("loss_dataset3")
    m = _model("dataset3")
    sig = m.loss_function.draw_sigma(t(g["data"]).cuda(), time=torch.from_numpy(g["time"]).cuda())
    loss = m.loss_function.loss_function(m, t(g["data"]).cuda(), t(g["E"]).cuda(), sigma=sig, noise=t(g["noise"]).cuda(), layers=None)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))


def test_philox_stream_properties():
    from calodiffusion_amd.engine import randn
    a = randn((1 << 20,), "cuda", seed=1234, offset=0)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.var()) - 1.0) < 5e-3
    k = float(((a - a.mean()) ** 4).mean() / a.var() ** 2)
    assert abs(k - 3.0) < 0.05
    # element i depends only on (seed, offset + i): shards of one stream agree with the whole
    b = randn((1000,), "cuda", seed=1234, offset=777)
    assert torch.equal(b, a[777:1777])
    assert torch.equal(randn((1 << 20,), "cuda", seed=1234, offset=0), a)
    assert not torch.equal(randn((1000,), "cuda", seed=1235, offset=0), a[:1000])


def test_full_size_properties_dataset2_batch64():
    """BASELINE's headline size (B = 64): size-independent properties instead of a 64-sample oracle run."""
    m = _model("dataset2")
    gen = torch.Generator().manual_seed(11)
    x = torch.randn((64, 1, 45, 16, 9), generator=gen).cuda()
    E = torch.rand((64, 1), generator=gen).cuda()
    layers = torch.randn((64, 46), generator=gen).cuda()
    sig = torch.full((64,), 1.3, device="cuda")
    y = m.denoise(x, E=E, sigma=sig, layers=layers)
    assert torch.isfinite(y).all()
    # (1) batch independence: no op mixes showers
    y8 = m.denoise(x[5:13].contiguous(), E=E[5:13].contiguous(), sigma=sig[5:13].contiguous(), layers=layers[5:13].contiguous())
    assert rel_l2(y8.cpu().numpy(), y[5:13].cpu().numpy()) < 2e-6
    # (2) phi periodicity: Dataset-2 has no phi input channel, so rolling the input along phi by a multiple of 4
    #     (two stride-2 levels) rolls the output
    yr = m.denoise(torch.roll(x, 4, dims=3).contiguous(), E=E, sigma=sig, layers=layers)
    assert rel_l2(torch.roll(yr, -4, dims=3).cpu().numpy(), y.cpu().numpy()) < 2e-6
    yr = m.denoise(torch.roll(x, 5, dims=3).contiguous(), E=E, sigma=sig, layers=layers)
    assert rel_l2(torch.roll(yr, -5, dims=3).cpu().numpy(), y.cpu().numpy()) > 1e-3  # odd shifts are NOT a symmetry
    # (3) determinism
    assert torch.equal(m.denoise(x, E=E, sigma=sig, layers=layers), y)
    # (4) agreement with the oracle on two of the 64 showers
    from oracle import torch_oracle as O
    om = O.OracleModel(m.config, {k: v.cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        want = om.denoise(x[[0, 63]].cpu(), E[[0, 63]].cpu(), sig[[0, 63]].cpu(), layers[[0, 63]].cpu())
    assert rel_l2(y[[0, 63]].cpu().numpy(), want.numpy()) < TOL_OP


def test_weight_update_is_picked_up():
    m = _model("tiny")
    g = gold("model_tiny")
    x, E, layers = t(g["x"]).cuda(), t(g["E"]).cuda(), t(g["layers"]).cuda()
    sig = torch.full((4,), 1.0, device="cuda")
    y0 = m.denoise(x, E=E, sigma=sig, layers=layers)
    with torch.no_grad():
        m.model.final_conv[1].conv.bias.add_(0.5)
    y1 = m.denoise(x, E=E, sigma=sig, layers=layers)
    c_out = 1.0 * 1.0 / np.sqrt(2.0)
    assert rel_l2((y1 - y0).cpu().numpy(), np.full(tuple(x.shape), 0.5 * c_out, dtype=np.float32)) < 1e-5
