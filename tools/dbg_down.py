import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from calodiffusion_amd.engine import Ops
from oracle import torch_oracle as O
ops = Ops()
gen = torch.Generator().manual_seed(1)
for shape, stride in (((1, 9, 8, 9), (2,2,2)), ((1,4,4,4),(2,2,2)), ((1,5,4,4),(2,2,2)), ((1,4,4,5),(2,2,2)), ((1,4,6,4),(1,2,2))):
    cin = cout = 32
    x = torch.randn((shape[0], cin) + shape[1:], generator=gen, requires_grad=True)
    w = (torch.randn((cout, cin, 3, 4, 4), generator=gen) * 0.1).requires_grad_()
    y = O.cyl_conv3d(x, w, None, stride=stride, padding=(1, 1, 1))
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    dx, dw, db = ops.conv_backward(ops.to_channels_last(x.detach().cuda()), w.detach().cuda(), ops.to_channels_last(dy.cuda()), stride=stride)
    got = ops.to_ncdhw(dx).cpu()
    err = (got - x.grad).norm() / x.grad.norm()
    # per-axis error profile
    d = (got - x.grad).abs().sum(dim=(0, 1))
    print(shape, stride, "rel err", float(err), "dw err", float((dw.cpu() - w.grad).norm() / w.grad.norm()))
    print("  err by z", d.sum(dim=(1, 2)).numpy().round(2))
    print("  err by h", d.sum(dim=(0, 2)).numpy().round(2))
    print("  err by w", d.sum(dim=(0, 1)).numpy().round(2))
