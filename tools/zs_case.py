"""z-slide conv op against the oracle for one (B, c0, c1, cout, D, H, W): python tools/zs_case.py 32 32 0 32 45 50 18"""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from calodiffusion_amd.engine import Ops
from oracle import torch_oracle as O
B, c0, c1, cout, D, H, W = (int(v) for v in sys.argv[1:8])
ops = Ops()
gen = torch.Generator().manual_seed(11)
cin = c0 + c1
x = torch.randn((B, cin, D, H, W), generator=gen)
w, bias = torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.05, torch.randn(cout, generator=gen)
want = O.cyl_conv3d(x, w, bias, padding=(1, 1, 1)).numpy()
xc = ops.to_channels_last(x.cuda())
y = ops.to_ncdhw(ops.cyl_conv(xc, w.cuda(), bias.cuda())).cpu().numpy()
err = np.sqrt(((y - want) ** 2).sum(axis=(1, 2, 3, 4)) / (want ** 2).sum(axis=(1, 2, 3, 4)))
print("per-sample rel err max", err.max(), "nan:", int(np.isnan(y).sum()), "bad samples:", np.nonzero(~(err < 2e-6))[0][:10])
if np.isnan(y).any() or not (err < 2e-6).all():
    bad = np.argwhere(~(np.abs(y - want) < 1e-3 * (1 + np.abs(want))))
    print("first bad (b,c,z,h,w):", bad[:5].tolist(), "count", len(bad))
    zs = np.unique(bad[:, 2]); print("bad z planes:", zs[:40])
    hs = np.unique(bad[:, 3]); print("bad phi rows:", hs[:60])
