"""Repeat short sampling runs and one training step many times; every repetition must be bitwise identical (race detector)."""
import sys
import torch
sys.path.insert(0, ".")
from calodiffusion_amd.calodiffusion import CaloDiffusion
from calodiffusion_amd.configs import load_config
name, B, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = load_config(name)
torch.manual_seed(1234)
m = CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
g = torch.Generator().manual_seed(1)
shape = [B] + list(cfg["SHAPE_PAD"][1:])
start = torch.randn(shape, generator=g).cuda()
E = torch.rand((B, 3 if cfg.get("HGCAL") else 1), generator=g).cuda()
layers = torch.randn((B, 1 + cfg["SHAPE_FINAL"][2]), generator=g).cuda() if "layer" in cfg["SHOWERMAP"] else None
ref = None
bad = 0
for r in range(reps):
    m.noise_offset = 0  # stochastic samplers draw from the running Philox offset
    out = m.sample(E, layers, num_steps=6, start=start)
    t = torch.from_numpy(out)
    if not torch.isfinite(t).all():
        print("non-finite at rep", r); bad += 1
    if ref is None:
        ref = t
    elif not torch.equal(ref, t):
        d = (ref - t).abs()
        print("MISMATCH rep", r, "max abs", float(d.max()), "count", int((d > 0).sum())); bad += 1
print(name, "B", B, "sampling reps", reps, "bad", bad)
if name != "hgcal":
    data, noise = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
    rnd = torch.randn((B,), generator=g).cuda()
    refg = None
    badt = 0
    for r in range(max(3, reps // 4)):
        m.zero_grad()
        kw = dict(rnd_normal=rnd) if "log" in cfg.get("NOISE_SCHED", "") else dict(time=torch.full((B,), 137).cuda())
        torch.manual_seed(5)
        loss = m.compute_loss(data, E, noise=noise, layers=layers, **kw)
        loss.backward()
        flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        if refg is None:
            refg = flat.clone()
        elif not torch.equal(refg, flat):
            print("TRAIN MISMATCH rep", r, float((refg - flat).abs().max())); badt += 1
    print(name, "train reps bad", badt)
