#!/usr/bin/env python3
"""Per-kernel breakdown of one training step (HIP events around every launch, cd_profile_*)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calodiffusion_amd.calodiffusion import CaloDiffusion
from calodiffusion_amd.configs import load_config
from calodiffusion_amd import engine

name = sys.argv[1] if len(sys.argv) > 1 else "dataset2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = load_config(name)
torch.manual_seed(1234)
m = CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
g = torch.Generator().manual_seed(1)
shape = [B] + list(cfg["SHAPE_PAD"][1:])
data, noise = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
E = torch.rand((B, 3 if cfg.get("HGCAL") else 1), generator=g).cuda()
layers = torch.randn((B, 1 + cfg["SHAPE_FINAL"][2]), generator=g).cuda() if "layer" in cfg["SHOWERMAP"] else None
rnd = torch.randn((B,), generator=g).cuda()
for _ in range(2):
    m.zero_grad(); m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd).backward()
torch.cuda.synchronize()
engine.profile_begin()
m.zero_grad(); m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd).backward()
prof = engine.profile_end()
tot = sum(v["ms"] for v in prof.values())
print(f"profiled kernels total {tot:.3f} ms (untimed small kernels excluded)")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:30]:
    tf = v["flops"] / (v["ms"] / v["launches"] * 1e-3) / 1e12 if v["flops"] else 0
    print(f"{k:44s} {v['ms']:8.3f} ms x{v['launches']:3d}  {tf:7.1f} TF")
