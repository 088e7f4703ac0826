"""GPU (MI355X): round-4 additions -- the noise_pred / mean_pred objectives behind cd_denoise / cd_loss_hybrid / cd_train_step
against the reference's own outputs, and the C ABI's independence of HIP errors left behind by earlier calls."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import gold, rel_l2
from helpers import t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("obj", ["noise_pred", "mean_pred"])
def test_objectives_match_the_reference(obj):
    """TRAINING_OBJ noise_pred / mean_pred: cd_denoise objectives 1 / 2 (calodiffusion.py:161-165) at three noise levels, a 6-step
    DDIM trajectory on that denoiser, and the loss classes of models/loss.py:181-210 -- value (cd_loss_hybrid) and every
    gradient (cd_train_step) for LOSS_TYPE l2 and huber -- against the reference (tests/golden/objectives_tiny.npz)."""
    from test_gpu_round2 import _model
    g = gold("objectives_tiny")
    data, E, noise, layers = (t(g[k]).cuda() for k in ("data", "E", "noise", "layers"))
    rnd = t(g["rnd_normal"]).cuda()
    m = _model("tiny", {"TRAINING_OBJ": obj, "LOSS_TYPE": "l2", "SAMPLER": "DDim"})  # (the tiny config's own sampler is DDPM)
    assert type(m.loss_function).__name__ == obj and type(m.sampler_algorithm).__name__ == "DDim"
    with torch.no_grad():
        for i, sg in enumerate(g["sigmas"]):
            xin = data * float(np.sqrt(1.0 + float(sg) ** 2))
            got = m.denoise(xin, E=E, sigma=torch.full((4,), float(sg)).cuda(), layers=layers)
            err = rel_l2(got.cpu().numpy(), g[f"{obj}.denoise_{i}"])
            assert err < 1e-5, (obj, i, err)
        x = m.sample(E, layers, num_steps=6, start=data)
        assert rel_l2(x, g[f"{obj}.ddim_6"]) < 1e-4
    for lt in ("l2", "huber"):
        m = _model("tiny", {"TRAINING_OBJ": obj, "LOSS_TYPE": lt})
        m.zero_grad()
        loss = m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)
        loss.backward()
        want = float(g[f"{obj}.{lt}.loss"])
        assert abs(float(loss) - want) <= 1e-5 * abs(want), (obj, lt, float(loss), want)
        with torch.no_grad():
            assert abs(float(m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)) - want) <= 1e-5 * abs(want)
        grads = dict(m.model.named_parameters())
        pre, worst = f"{obj}.{lt}.grad.", 0.0
        for k in g.files:
            if k.startswith(pre):
                err = rel_l2(grads[k[len(pre):]].grad.cpu().numpy(), g[k])
                worst = max(worst, err)
                assert err < 1e-4, (k, err)
        for k, (s1, s2) in zip(g[f"{obj}.{lt}.ck_keys"], g[f"{obj}.{lt}.ck_vals"]):
            gr = grads[str(k)].grad.double()
            assert abs(float((gr * gr).sum()) - s2) <= 2e-4 * max(s2, 1e-30), (obj, lt, k)
        print(f"[{obj}/{lt}] loss {float(loss):.6f} (reference {want:.6f}); worst whole-tensor gradient error {worst:.2e}")


def test_a_stale_hip_error_does_not_fail_the_next_entry_point():
    """The launchers check hipGetLastError() after every launch, which reports the thread's last error whoever caused it (round 3:
    a refused hipEventElapsedTime made the next cd_randn fail with 'invalid resource handle').  Every C-ABI entry point now
    clears the error state on the way in: provoke a HIP error outside the library, then call into it."""
    from calodiffusion_amd import engine
    lib = engine.load_library()
    torch.cuda.init()
    # the HIP runtime this process already runs on (torch ships its own copy: a second one would keep its own error state)
    path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln)
    hip = ctypes.CDLL(path)
    hip.hipFree.argtypes, hip.hipFree.restype = [ctypes.c_void_p], ctypes.c_int
    hip.hipPeekAtLastError.restype = ctypes.c_int
    torch.cuda.synchronize()
    assert hip.hipFree(ctypes.c_void_p(0x1234)) != 0      # not an allocation: fails and leaves the error behind
    assert hip.hipPeekAtLastError() != 0
    out = torch.empty(4096, dtype=torch.float32, device="cuda")
    rc = lib.cd_randn(out.data_ptr(), out.numel(), 7, 0, engine._stream())
    assert rc == 0, lib.cd_last_error().decode(errors="replace")
    torch.cuda.synchronize()
    assert abs(float(out.mean())) < 0.1 and 0.9 < float(out.std()) < 1.1
    # ... and a failing entry point leaves nothing behind for the next one either
    assert lib.cd_randn(None, 16, 7, 0, engine._stream()) != 0
    assert lib.cd_randn(out.data_ptr(), out.numel(), 7, 0, engine._stream()) == 0


def test_full_resolution_conv_writes_nothing_outside_its_output():
    """Guard bands around the output of the z-slide convolution at the chunk lengths the sampling loop runs (the parity tests
    compare what lies INSIDE the output): long chunks of whole planes (Dataset-2 at batch 64: 4 chunks of 26 steps per sample), the
    continuation launch of a 64-channel input, phi strips with a 5-plane ring (Dataset-3, HGCal).  Written after round 3's
    unexplained core dump: every store of the shipped instances -- the row stores of the reduction and the
    partial-statistics stores -- stays inside its tensor."""
    import ctypes as C
    from calodiffusion_amd import engine
    ops = engine.Ops()
    gen = torch.Generator().manual_seed(17)
    pad = 1 << 16  # floats either side
    for B, cin, cout, shape in ((64, 32, 32, (45, 16, 9)), (64, 64, 32, (45, 16, 9)), (8, 32, 32, (45, 50, 18)), (16, 32, 32, (28, 12, 21)),
                                (3, 32, 64, (7, 16, 8))):
        D, H, W = shape
        x = torch.randn((B, D, H, W, cin), generator=gen).cuda()
        w = (torch.randn((cout, cin, 3, 3, 3), generator=gen) / (27 * cin) ** 0.5).cuda()
        bias = torch.randn((cout,), generator=gen).cuda()
        n = B * D * H * W * cout
        big = torch.full((pad + n + pad,), 7.25, dtype=torch.float32, device="cuda")
        sc = ops.scratch(B, max(cin, cout), D * H * W)
        engine._check(ops.lib.cd_op_cyl_conv(x.data_ptr(), cin, None, 0, w.data_ptr(), bias.data_ptr(), big.data_ptr() + 4 * pad, B,
                                             cout, engine._i32x3((D, H, W)), engine._i32x3((3, 3, 3)), engine._i32x3((1, 1, 1)),
                                             sc.data_ptr(), engine._stream()))
        torch.cuda.synchronize()
        assert bool((big[:pad] == 7.25).all()) and bool((big[pad + n:] == 7.25).all()), (B, cin, cout, shape)
        y = big[pad:pad + n].view(B, D, H, W, cout)
        assert bool(torch.isfinite(y).all()) and not bool((y == 7.25).any()), (B, cin, cout, shape)  # every element written
        ref = ops.cyl_conv(x, w, bias)
        assert torch.equal(ref, y)  # (deterministic: the same launch into an ordinary tensor)


def test_workspace_cache_follows_precision_and_switches(monkeypatch):
    """A cached workspace stays valid when the arithmetic mode changes after the first call (the library sizes for the largest of
    its modes), a launch-sequence switch gets its own workspace, and at most one network / one sampler workspace is kept (ADVICE
    r03: a ragged last batch doubled the footprint, a precision switch after the first call failed with CD_EWORKSPACE)."""
    from calodiffusion_amd import engine
    from test_gpu_round2 import _model
    m = _model("dataset2", {"SAMPLER": "Heun"})
    eng = m.engine()
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 1, 45, 16, 9), generator=g).cuda()
    E, layers = torch.rand((2, 1), generator=g).cuda(), torch.randn((2, 46), generator=g).cuda()
    sig = torch.tensor([2.0, 0.3]).cuda()
    ref = m.denoise(x, E=E, sigma=sig, layers=layers)
    try:
        for mode in ("f32", "bf16x3"):
            engine.set_conv_precision(mode)
            y = m.denoise(x, E=E, sigma=sig, layers=layers)
            assert rel_l2(y.cpu().numpy(), ref.cpu().numpy()) < 1e-5, mode
    finally:
        engine.set_conv_precision("f16x2")
    monkeypatch.setenv("CD_NO_DEEP_LEVEL", "1")
    y = m.denoise(x, E=E, sigma=sig, layers=layers)
    assert rel_l2(y.cpu().numpy(), ref.cpu().numpy()) < 3e-6
    monkeypatch.delenv("CD_NO_DEEP_LEVEL")
    m.denoise(x[:1], E=E[:1], sigma=sig[:1], layers=layers[:1])  # another batch size: the previous workspace is evicted
    m.sample(E, layers, num_steps=3, start=x)
    m.sample(E[:1], layers[:1], num_steps=3, start=x[:1])
    kinds = [k[0] for k in eng._ws]
    assert kinds.count("net") == 1 and kinds.count("sampler") == 1, list(eng._ws)


@pytest.mark.parametrize("parts", [2, 4])
def test_cooperating_workgroups_attention_equals_the_single_workgroup_form(parts):
    """attn_coop_kernel (CD_ATTN_COOP: a sample's voxels dealt to several co-operating workgroups with two in-launch barriers)
    against the one-workgroup-per-sample attn_small_kernel at Dataset-2's level-1 grids (736 voxels, 32 and 64 channels, batch a
    multiple of 8 so that a sample's workgroups share an XCD) -- run in a child process, CD_ATTN_COOP is read once.  The two forms
    merge their {max, sum, context} partials differently, so they agree to rounding, not bit for bit; the range flag must stay
    clear (a barrier that times out raises it)."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_gpu_round2 import _model
m = _model("dataset2")
g = torch.Generator().manual_seed(5)
B = 16
x = torch.randn((B, 1, 45, 16, 9), generator=g).cuda()
E, layers = torch.rand((B, 1), generator=g).cuda(), torch.randn((B, 46), generator=g).cuda()
sig = torch.linspace(0.05, 30.0, B).cuda()
eng = m.engine()
eng.safe_denoise = False
y = m.denoise(x, E=E, sigma=sig, layers=layers)
torch.cuda.synchronize()
eng.check_status()
np.save(sys.argv[1], y.cpu().numpy())
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, env in (("single", {}), ("coop", {"CD_ATTN_COOP": str(parts)})):
        path = f"/tmp/coop_{parts}_{tag}.npy"
        e = dict(os.environ, **env)
        subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "tests")), path], check=True, env=e, timeout=600)
        outs[tag] = np.load(path)
    err = rel_l2(outs["coop"], outs["single"])
    print(f"co-operative attention ({parts} workgroups per sample) vs single workgroup: rel L2 {err:.2e}")
    assert err < 3e-6


@pytest.mark.parametrize("shape,batch", [((45, 16, 9), 4), ((28, 12, 21), 3), ((45, 50, 18), 2)])
def test_attention_moment_form_equals_the_separate_closing_pass(shape, batch, monkeypatch):
    """Level-0 attention (32 channels, > 1024 voxels): the moment form -- pass 1 also accumulates sum(softmax(q) - 1/32) and its
    32 x 32 second moment, pass 2 derives the closing GroupNorm(1)'s mean / variance from them and the folded weights and writes
    gn(y) + x directly -- beside the three-launch form (y written, channel sums, gn_apply), both against the oracle in float64, on grids with a ragged last tile,
    for near-uniform softmaxes (default-init scale), peaked ones (q weights x 12) and an output projection with a large bias
    (mean^2 >> variance: the cancellation case of E[y^2] - mean^2)."""
    import os
    from calodiffusion_amd import engine
    if os.environ.get("CD_CONV_PRECISION", "f16x2") != "f16x2":
        pytest.skip("the fused attention passes (and with them the moment form) run in the default f16x2 mode only")
    monkeypatch.setenv("CD_ATTN_MOM_MIN", "0")  # (the plan takes this form from 4 M elements per tensor)
    ops = engine.Ops()
    gen = torch.Generator().manual_seed(23)
    D, H, W = shape
    C = 32
    x = (torch.randn((batch, D, H, W, C), generator=gen) * 1.5 + 0.3).cuda()
    for qscale, bias in ((1.0, 0.0), (12.0, 0.0), (1.0, 3.0)):
        wqkv = torch.randn((96, C, 1, 1, 1), generator=gen) / C ** 0.5
        wqkv[:32] *= qscale
        sd = {"fn.norm.weight": 1 + 0.1 * torch.randn(C, generator=gen), "fn.norm.bias": 0.1 * torch.randn(C, generator=gen),
              "fn.fn.to_qkv.conv.weight": wqkv, "fn.fn.to_out.0.conv.weight": torch.randn((C, 32, 1, 1, 1), generator=gen) / 32 ** 0.5,
              "fn.fn.to_out.0.conv.bias": 0.1 * torch.randn(C, generator=gen) + bias,
              "fn.fn.to_out.1.weight": 1 + 0.1 * torch.randn(C, generator=gen), "fn.fn.to_out.1.bias": 0.1 * torch.randn(C, generator=gen)}
        sd = {k: v.cuda().contiguous() for k, v in sd.items()}
        y_mom = ops.linear_attention(x, sd)
        monkeypatch.setenv("CD_NO_ATTN_MOMENTS", "1")
        y_sep = ops.linear_attention(x, sd)
        monkeypatch.delenv("CD_NO_ATTN_MOMENTS")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y_mom).all())
        assert not torch.equal(y_mom, y_sep)  # (the two forms really are different launches)
        # both against the oracle in float64, on the attention branch alone (the residual x is common and larger than it)
        from oracle import torch_oracle
        sd64 = {"a." + k: v.double().cpu() for k, v in sd.items()}
        x64 = x.double().cpu().permute(0, 4, 1, 2, 3)
        want = (torch_oracle.attn_residual(sd64, "a", x64, True) - x64).permute(0, 2, 3, 4, 1).numpy()
        e_mom = rel_l2((y_mom.double().cpu() - x.double().cpu()).numpy(), want)
        e_sep = rel_l2((y_sep.double().cpu() - x.double().cpu()).numpy(), want)
        print(f"{shape} q x{qscale} bias {bias}: rel L2 of the branch against float64 -- moment form {e_mom:.2e}, closing pass {e_sep:.2e}")
        assert e_mom < 2e-6, (shape, qscale, bias, e_mom, e_sep)


def _sde_cases():
    from test_host import SDE_CASES
    return SDE_CASES


@pytest.mark.parametrize("name,opts", _sde_cases())
def test_sde_samplers_on_the_device_equal_the_restated_loops(name, opts):
    """DPMPPSDE / DPMPP2MSDE / DPMPP3MSDE (models/sample.py:347-574) through cd_sampler_run with the unit normals injected, against
    the reference's loops restated on the CPU oracle with the same normals (torchsde is absent: parity unpinned, there is no
    reference trajectory for these classes); and, drawing from the device Philox stream, two runs differ while ETA = 0 is
    deterministic."""
    from oracle import samplers_oracle as SO
    from oracle import torch_oracle as O
    from test_gpu_round2 import _model
    from test_host import sde_oracle_run
    m = _model("tiny", {"SAMPLER": name, "SAMPLER_OPTIONS": dict(opts)})
    smp = m.sampler_algorithm
    assert type(smp).__name__ == name
    n, rows = 7, 2
    gen = torch.Generator().manual_seed(43)
    start = torch.randn((rows, 1, 8, 8, 8), generator=gen)
    E, layers = torch.rand((rows, 3), generator=gen), torch.randn((rows, 9), generator=gen)
    prog = smp.build(m, n, 0).finalize()
    sig = SO.model_sigmas(m.loss_function, n)
    m.loss_function.update_step(m.nsteps)
    noise = [torch.randn(start.shape, generator=gen) for _ in range(prog.n_randn)]
    om = O.OracleModel(m.config, {k[6:]: v.detach().cpu() for k, v in m.state_dict().items()})
    den = lambda x, s: om.denoise(x, E, torch.as_tensor(s).float().expand(rows), layers)  # noqa: E731
    with torch.no_grad():
        want = sde_oracle_run(name, opts, den, start, sig, noise)
    if prog.n_randn:
        smp.step_noise = torch.stack(noise).cuda()
    got = m.sample(E.cuda(), layers.cuda(), num_steps=n, start=start.cuda())
    err = rel_l2(np.asarray(got), want.numpy())
    assert err < 1e-4, (name, opts, err)
    # the device's own stream
    smp.step_noise = None
    a = m.sample(E.cuda(), layers.cuda(), num_steps=n, start=start.cuda())
    b = m.sample(E.cuda(), layers.cuda(), num_steps=n, start=start.cuda())
    a, b = np.asarray(a), np.asarray(b)
    assert np.isfinite(a).all() and np.isfinite(b).all()
    if opts.get("ETA"):
        assert rel_l2(a, b) > 1e-3  # fresh noise per call
    else:
        assert np.array_equal(a, b) and rel_l2(a, want.numpy()) < 1e-4
