#!/usr/bin/env python3
"""Every launch category of one eager denoise step with its HIP-event time (the library's own profiler): which kernel, at which
geometry, how many launches, how long.

    python tools/step_breakdown.py [--config dataset2 --batch 64]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="dataset2")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
from calodiffusion_amd import engine  # noqa: E402
from calodiffusion_amd.calodiffusion import CaloDiffusion  # noqa: E402
from calodiffusion_amd.configs import load_config  # noqa: E402

cfg = load_config(a.config)
torch.manual_seed(1234)
m = CaloDiffusion(cfg, n_steps=400, loss_type="l2")
B = a.batch
g = torch.Generator().manual_seed(1)
x = torch.randn([B] + list(cfg["SHAPE_PAD"][1:]), generator=g).cuda()
n_e = 3 if cfg.get("HGCAL", False) else 1
E = torch.rand((B, n_e), generator=g).cuda()
layers = torch.randn((B, cfg["SHAPE_PAD"][2] + 1), generator=g).cuda() if "layer" in cfg.get("SHOWERMAP", "") else None
sig = torch.full((B,), 1.5).cuda()
for _ in range(3):
    m.denoise(x, E=E, sigma=sig, layers=layers)
torch.cuda.synchronize()
engine.profile_begin()
for _ in range(a.reps):
    m.denoise(x, E=E, sigma=sig, layers=layers)
prof = engine.profile_end()
rows = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows) / a.reps
print(f"{a.config} batch {B}: {sum(v['launches'] for _, v in rows) // a.reps} launches, {tot * 1e3:.1f} us of kernels per eager step")
for k, v in rows:
    n = v["launches"] / a.reps
    us = v["ms"] * 1e3 / a.reps
    extra = ""
    if v.get("flops"):
        extra = f"  {v['flops'] / (us / n * 1e-6) / 1e12:7.1f} TFLOP/s"
    if v.get("bytes"):
        extra += f"  {v['bytes'] / (us / n * 1e-6) / 1e12:6.2f} TB/s"
    print(f"{us:8.1f} us  {n:5.1f} x {us / n:7.1f} us  {k}{extra}")
