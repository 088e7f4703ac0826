#!/usr/bin/env python3
"""Micro-benchmark of the 3x3x3 weight gradient through the C ABI (cd_op_conv_backward without dx / db), event-timed back to back.

    python tools/wgrad_bench.py [--batch 32 --cin 32 --cout 32 --dims 45,16,9 --iters 20]
Experiment builds (-DCD_WGRAD_ABL) read CD_WGRAD_ABL: 1 no MFMAs, 2 no K loop, 4 no prefetch of the next unit, 8 no partial write."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calodiffusion_amd.engine import Ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--dims", default="45,16,9")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    D, H, W = (int(v) for v in a.dims.split(","))
    ops = Ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn((a.batch, D, H, W, a.cin), generator=g).cuda()
    dy = torch.randn((a.batch, D, H, W, a.cout), generator=g).cuda()
    w = (torch.randn((a.cout, a.cin, 3, 3, 3), generator=g) * 0.05).cuda()
    for _ in range(3):
        _, dw, _ = ops.conv_backward(x, w, dy, need_dx=False, bias=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        _, dw, _ = ops.conv_backward(x, w, dy, need_dx=False, bias=False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3  # absmax pass + weight-gradient kernel + slot reduction
    print(f"abl={os.environ.get('CD_WGRAD_ABL', '0'):>2s} ring={'off' if os.environ.get('CD_NO_WGRAD_RING') else 'on'} "
          f"B{a.batch} {a.cin}->{a.cout} @{a.dims}: {us:8.1f} us per call  checksum {float(dw.double().sum()):.6e}")


if __name__ == "__main__":
    main()
