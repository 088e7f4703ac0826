import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from calodiffusion_amd.engine import Ops
from oracle import torch_oracle as O
ops = Ops()
gen = torch.Generator().manual_seed(1)
B, cin, cout, shape = 1, 32, 32, (4, 16, 9)
x = torch.randn((B, cin) + shape, generator=gen)
def run(w, tag):
    want = O.cyl_conv3d(x, w, None, padding=(1, 1, 1))
    y = ops.to_ncdhw(ops.cyl_conv(ops.to_channels_last(x.cuda()), w.cuda(), None)).cpu()
    err = float((y - want).norm() / want.norm())
    print(tag, "rel err", err)
    return y, want
w = torch.zeros(cout, cin, 3, 3, 3); 
for c in range(32): w[c, c, 1, 1, 1] = 1.0
y, want = run(w, "identity")
if (y - want).abs().max() > 1e-4:
    print(" y[0,:4,0,0,:4]", y[0, :4, 0, 0, :4].numpy().round(3)); print(" want", want[0, :4, 0, 0, :4].numpy().round(3))
    # find which input element each output matches
    yy = y[0, 0].flatten()[:20]; xx = x[0].flatten()
    for v in yy[:6]:
        idx = (xx - v).abs().argmin(); print("  out", float(v), "closest x idx", np.unravel_index(int(idx), x[0].shape), float(xx[idx]))
for tap in ((0, 1, 1), (2, 1, 1), (1, 0, 1), (1, 2, 1), (1, 1, 0), (1, 1, 2)):
    w = torch.zeros(cout, cin, 3, 3, 3)
    for c in range(32): w[c, c][tap] = 1.0
    run(w, f"shift {tap}")
w = torch.zeros(cout, cin, 3, 3, 3); w[:, :, 1, 1, 1] = torch.randn(32, 32, generator=gen)
run(w, "center mix")
w = torch.randn(cout, cin, 3, 3, 3, generator=gen) * 0.05
run(w, "full")
