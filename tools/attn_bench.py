#!/usr/bin/env python3
"""Micro-benchmark of Residual(PreNorm(LinearAttention)) through the C ABI (cd_op_linear_attention): per-kernel time and HBM
fraction of the two fused passes, at several sequence lengths (fixed vs per-tile cost).

    python tools/attn_bench.py [--batch 64 --channels 32 --dims 45,16,9 --iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calodiffusion_amd import engine  # noqa: E402
from calodiffusion_amd.engine import Ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--dims", default="45,16,9")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    D, H, W = (int(v) for v in a.dims.split(","))
    C = a.channels
    ops = Ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn((a.batch, D, H, W, C), generator=g).cuda()
    sd = {
        "fn.norm.weight": 1 + 0.1 * torch.randn(C, generator=g), "fn.norm.bias": 0.1 * torch.randn(C, generator=g),
        "fn.fn.to_qkv.conv.weight": 0.2 * torch.randn(96, C, 1, 1, 1, generator=g),
        "fn.fn.to_out.0.conv.weight": 0.2 * torch.randn(C, 32, 1, 1, 1, generator=g),
        "fn.fn.to_out.0.conv.bias": 0.1 * torch.randn(C, generator=g),
        "fn.fn.to_out.1.weight": 1 + 0.1 * torch.randn(C, generator=g), "fn.fn.to_out.1.bias": 0.1 * torch.randn(C, generator=g),
    }
    sd = {k: v.cuda().contiguous() for k, v in sd.items()}
    for _ in range(3):
        y = ops.linear_attention(x, sd)
    torch.cuda.synchronize()
    engine.profile_begin()
    for _ in range(a.iters):
        y = ops.linear_attention(x, sd)
    prof = engine.profile_end()
    nbytes = x.numel() * 4
    for k, v in prof.items():
        if not v["launches"]:
            continue
        us = v["ms"] / v["launches"] * 1e3
        passes = {"attn_kv_context": 1, "attn_out": 2, "gn_apply": 3}.get(k.split(" ")[0])
        extra = f"  {passes * nbytes / us / 1e6:6.2f} TB/s ({passes * nbytes / us / 1e6 / 8 * 100:4.1f}% of 8 TB/s)" if passes else ""
        print(f"{k:24s} {us:8.1f} us{extra}")
    print(f"checksum {float(y.double().sum()):.6e}")


if __name__ == "__main__":
    main()
