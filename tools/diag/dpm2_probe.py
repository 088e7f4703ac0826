"""Where the dpm_2 case's device error comes from: denoise error at its two evaluations, and the step program interpreted on the
CPU with the DEVICE denoiser (separates the LINDIV kernel from the denoise kernels).  GPU box only."""
import copy, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import gold, rel_l2
from helpers import seeded_unet, t
from sampler_cases import CASES
from test_host import _interpret_program
from oracle import torch_oracle as O
from calodiffusion_amd.calodiffusion import CaloDiffusion
from calodiffusion_amd.configs import load_config

g = gold("samplers_tiny")
tag = "dpm_2"
name, over, _, off, rows = CASES[tag]
cfg = copy.deepcopy(load_config("tiny")); cfg.update(over); cfg["SAMPLER"] = name
torch.manual_seed(1234)
m = CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
om = O.OracleModel(cfg, {k: v.cpu() for k, v in m.state_dict().items()})
smp = m.sampler_algorithm
n = int(g[f"{tag}.n"])
prog = smp.build(m, n, off).finalize()
start, E, layers = t(g["start"])[:rows], t(g["E"])[:rows], t(g["layers"])[:rows]
calls = []
def den_o(x, s):
    y = om.denoise(x, E, s.float().expand(rows), layers)
    return y
def den_d(x, s):
    y = m.denoise(x.cuda(), E=E.cuda(), sigma=s.float().expand(rows).cuda(), layers=layers.cuda()).cpu()
    yo = om.denoise(x, E, s.float().expand(rows), layers)
    calls.append((float(s), rel_l2(y.numpy(), yo.numpy()), float(yo.norm()), float(x.norm())))
    return y
with torch.no_grad():
    xo = _interpret_program(prog, den_o, start, [])[0]
    xd = _interpret_program(prog, den_d, start, [])[0]
print("oracle-interpreted vs golden", rel_l2(xo.numpy(), g[f"{tag}.x"]))
print("device-denoise-interpreted vs golden", rel_l2(xd.numpy(), g[f"{tag}.x"]))
for c in calls: print("sigma %.4g: device vs oracle denoise rel %.2e (|D| %.3g, |x| %.3g)" % c)
m.loss_function.update_step(m.nsteps)
x = m.sample(E.cuda(), layers.cuda(), num_steps=n, start=start.cuda(), sample_offset=off)
print("device program vs golden", rel_l2(np.asarray(x), g[f"{tag}.x"]), " vs device-denoise-interpreted", rel_l2(np.asarray(x), xd.numpy()))
for prec in ("bf16x3", "f32"):
    from calodiffusion_amd import engine
    engine.set_conv_precision(prec)
    x = m.sample(E.cuda(), layers.cuda(), num_steps=n, start=start.cuda(), sample_offset=off)
    print(prec, "device program vs golden", rel_l2(np.asarray(x), g[f"{tag}.x"]))
