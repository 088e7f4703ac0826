#!/usr/bin/env python3
"""Phase breakdown of the one-launch deepest level (kernels_deep.hip) from in-kernel s_memtime stamps.

    CD_BUILD_TAG=exp CD_EXTRA_HIPCC_FLAGS="-DCD_ZS_EXPERIMENTS -DCD_DEEP_STAMPS" python -m calodiffusion_amd.build
    CALODIFF_LIB=$PWD/calodiffusion_amd/lib/libcalodiff_hip_exp.so CD_DEEP_DBG=1 python tools/deep_stamps.py

Prints the stamp sums of workgroup 0 / wave 0 for one Dataset-2 denoise call at batch 64 (stderr lines "[deep stamps] ...") and the
launch's duration from HIP events for the tick -> microsecond conversion."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calodiffusion_amd import engine  # noqa: E402
from calodiffusion_amd.calodiffusion import CaloDiffusion  # noqa: E402
from calodiffusion_amd.configs import load_config  # noqa: E402

cfg = dict(load_config("dataset2"))
torch.manual_seed(1234)
m = CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
B = int(os.environ.get("B", 64))
g = torch.Generator().manual_seed(1)
x = torch.randn([B] + list(cfg["SHAPE_PAD"][1:]), generator=g).cuda()
E, layers = torch.rand((B, 1), generator=g).cuda(), torch.randn((B, 46), generator=g).cuda()
sig = torch.ones(B).cuda()
m.engine().safe_denoise = False
for _ in range(3):
    m.denoise(x, E=E, sigma=sig, layers=layers)
torch.cuda.synchronize()
engine.profile_begin()
for _ in range(3):
    m.denoise(x, E=E, sigma=sig, layers=layers)
prof = engine.profile_end()
for k, v in prof.items():
    if k.startswith("deep_level"):
        print(k, f"{v['ms'] / v['launches'] * 1e3:.1f} us per launch")
