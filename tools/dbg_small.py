"""Parity probe for the whole-sample-in-LDS conv (kernels_conv_small.hip) against the CPU oracle."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from calodiffusion_amd.engine import Ops
from oracle import torch_oracle as O
ops = Ops()
gen = torch.Generator().manual_seed(1)
for B, cin, cout, shape in ((2, 32, 32, (12, 4, 2)), (3, 64, 64, (12, 4, 2)), (2, 64, 32, (12, 4, 2)), (1, 128, 64, (5, 5, 5)), (2, 96, 32, (3, 1, 4)),
                            (64, 64, 64, (12, 4, 2)), (2, 32, 32, (1, 2, 3))):
    x = torch.randn((B, cin) + shape, generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.05
    bias = torch.randn(cout, generator=gen)
    nb = min(B, 3)
    want = O.cyl_conv3d(x[:nb], w, bias, padding=(1, 1, 1))
    if cin == 64 and cout == 32:
        y = ops.cyl_conv(ops.to_channels_last(x[:, :32].contiguous().cuda()), w.cuda(), bias.cuda(), x1_cl=ops.to_channels_last(x[:, 32:].contiguous().cuda()))
    else:
        y = ops.cyl_conv(ops.to_channels_last(x.cuda()), w.cuda(), bias.cuda())
    got = ops.to_ncdhw(y).cpu()[:nb]
    err = (got - want).norm() / want.norm()
    print(B, cin, cout, shape, "rel err", float(err))
