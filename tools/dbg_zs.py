"""Debug / parity probe for the z-slide f16x2 conv: error profile against the CPU oracle on Dataset-2's level-0 grid."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from calodiffusion_amd.engine import Ops
from oracle import torch_oracle as O
ops = Ops()
gen = torch.Generator().manual_seed(1)
for B, cin, cout, shape in ((1, 32, 32, (45, 16, 9)), (3, 32, 32, (45, 16, 9)), (2, 64, 32, (10, 16, 9)), (2, 32, 64, (7, 16, 8)), (64, 32, 32, (45, 16, 9))):
    x = torch.randn((B, cin) + shape, generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.05
    bias = torch.randn(cout, generator=gen)
    nb = min(B, 3)
    want = O.cyl_conv3d(x[:nb], w, bias, padding=(1, 1, 1))
    if cin == 64:
        y = ops.cyl_conv(ops.to_channels_last(x[:, :32].contiguous().cuda()), w.cuda(), bias.cuda(), x1_cl=ops.to_channels_last(x[:, 32:].contiguous().cuda()))
    else:
        y = ops.cyl_conv(ops.to_channels_last(x.cuda()), w.cuda(), bias.cuda())
    got = ops.to_ncdhw(y).cpu()[:nb]
    err = (got - want).norm() / want.norm()
    print(B, cin, cout, shape, "rel err", float(err))
    if err > 1e-5:
        d = (got - want).abs().sum(dim=(0, 1))
        print("  err by z", d.sum(dim=(1, 2)).numpy().round(1))
        print("  err by h", d.sum(dim=(0, 2)).numpy().round(1))
        print("  err by w", d.sum(dim=(0, 1)).numpy().round(1))
        print("  err by c", (got - want).abs().sum(dim=(0, 2, 3, 4)).numpy().round(1))
# phi strips: Dataset-3 (50x18 planes, strips of 5 rows) and HGCal (12x21, strips of 4) grids
for B, cin, cout, shape in ((1, 32, 32, (6, 50, 18)), (2, 32, 32, (9, 12, 21)), (1, 32, 32, (45, 50, 18)), (2, 64, 32, (5, 10, 18))):
    x = torch.randn((B, cin) + shape, generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.05
    bias = torch.randn(cout, generator=gen)
    want = O.cyl_conv3d(x, w, bias, padding=(1, 1, 1))
    y = ops.cyl_conv(ops.to_channels_last(x.cuda()), w.cuda(), bias.cuda())
    got = ops.to_ncdhw(y).cpu()
    err = (got - want).norm() / want.norm()
    print(B, cin, cout, shape, "rel err", float(err))
    if err > 1e-5:
        d = (got - want).abs().sum(dim=(0, 1))
        print("  err by z", d.sum(dim=(1, 2)).numpy().round(1))
        print("  err by h", d.sum(dim=(0, 2)).numpy().round(1))
        print("  err by w", d.sum(dim=(0, 1)).numpy().round(1))
