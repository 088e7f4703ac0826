#!/usr/bin/env python3
"""Micro-benchmark of the cylindrical 3x3x3 conv through the C ABI (cd_op_cyl_conv): algorithmic TFLOP/s vs fp32 MFMA peak.

    python tools/conv_bench.py [--batch 64 --cin 32 --cout 32 --dims 45,16,9 --iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calodiffusion_amd.engine import Ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--dims", default="45,16,9")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    D, H, W = (int(v) for v in a.dims.split(","))
    ops = Ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn((a.batch, D, H, W, a.cin), generator=g).cuda()
    w = (torch.randn((a.cout, a.cin, 3, 3, 3), generator=g) * 0.05).cuda()
    b = torch.randn((a.cout,), generator=g).cuda()
    for _ in range(3):
        y = ops.cyl_conv(x, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        y = ops.cyl_conv(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters  # includes the (tiny) weight-pack kernel of the op entry point
    flops = 2.0 * 27 * a.cin * a.cout * D * H * W * a.batch
    from calodiffusion_amd import engine
    engine.profile_begin()
    for _ in range(a.iters):
        y = ops.cyl_conv(x, w, b)
    prof = engine.profile_end()
    kern = {k: round(v["ms"] / v["launches"] * 1e3, 1) for k, v in prof.items() if v["launches"]}
    print("kernel us:", kern)
    print(f"tile={os.environ.get('CD_CONV_TILE','auto'):>12s}  {ms*1e3:8.1f} us  {flops/ms/1e9:7.2f} TFLOP/s  "
          f"({flops/ms/1e9/157.3*100:5.1f}% of fp32 MFMA peak)  checksum {float(y.double().sum()):.6e}")


if __name__ == "__main__":
    main()
