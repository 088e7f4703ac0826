"""GPU (MI355X): the backward kernels, through the C ABI, against torch autograd on the CPU oracle's ops.

The reference trains through torch autograd over stock modules (train/train_diffusion.py:52-63); the oracle restates
those modules functionally, so autograd through the oracle is the gradient oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    from calodiffusion_amd.engine import Ops
    return Ops()


def cl(ops, a):
    return ops.to_channels_last(a.detach().cuda())


def back(ops, y_cl):
    return ops.to_ncdhw(y_cl).cpu().numpy()


def _conv_case(ops, cin, cout, shape, k, stride, gen, split=None, dy_scale=1.0):
    x = torch.randn((shape[0], cin) + shape[1:], generator=gen, requires_grad=True)
    w = (torch.randn((cout, cin) + k, generator=gen) * 0.1).requires_grad_()
    b = torch.randn(cout, generator=gen, requires_grad=True)
    pad = (0, 0, 0) if k == (1, 1, 1) else (1, 1, 1)
    y = O.cyl_conv3d(x, w, b, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=gen) * dy_scale
    y.backward(dy)
    if split:
        x0, x1 = x[:, :split], x[:, split:]
        dx, dw, db = ops.conv_backward(cl(ops, x0), w.detach().cuda(), cl(ops, dy), stride=stride, x1_cl=cl(ops, x1))
    else:
        dx, dw, db = ops.conv_backward(cl(ops, x), w.detach().cuda(), cl(ops, dy), stride=stride)
    assert rel_l2(back(ops, dx), x.grad.numpy()) < TOL, ("dx", cin, cout, k)
    assert rel_l2(dw.cpu().numpy(), w.grad.numpy()) < TOL, ("dw", cin, cout, k)
    assert rel_l2(db.cpu().numpy(), b.grad.numpy()) < TOL, ("db", cin, cout, k)


def test_conv_backward(ops):
    gen = torch.Generator().manual_seed(21)
    _conv_case(ops, 32, 32, (2, 5, 6, 4), (3, 3, 3), (1, 1, 1), gen)
    _conv_case(ops, 64, 32, (1, 4, 3, 5), (3, 3, 3), (1, 1, 1), gen)
    _conv_case(ops, 128, 32, (1, 3, 4, 2), (3, 3, 3), (1, 1, 1), gen, split=64)   # skip-concat input
    _conv_case(ops, 96, 64, (1, 3, 5, 2), (3, 3, 3), (1, 1, 1), gen)
    _conv_case(ops, 64, 32, (2, 3, 4, 5), (1, 1, 1), (1, 1, 1), gen)
    _conv_case(ops, 128, 32, (1, 3, 4, 2), (1, 1, 1), (1, 1, 1), gen, split=64)
    _conv_case(ops, 32, 32, (1, 9, 8, 9), (3, 4, 4), (2, 2, 2), gen)             # Dataset-2-like down-sampling
    _conv_case(ops, 32, 32, (2, 5, 5, 7), (3, 4, 4), (2, 2, 2), gen)             # odd extents
    _conv_case(ops, 64, 64, (1, 4, 6, 4), (3, 4, 4), (1, 2, 2), gen)             # COMPRESS_Z false
    # planes of >= 64 voxels: the weight gradient runs on the fp16 matrix pipe (kernels_wgrad16.hip); odd plane counts,
    # concatenated input, and gradients far below the fp16 normal range (the kernel rescales them by a power of two)
    _conv_case(ops, 32, 32, (2, 5, 16, 9), (3, 3, 3), (1, 1, 1), gen)
    _conv_case(ops, 64, 32, (1, 3, 16, 9), (3, 3, 3), (1, 1, 1), gen, split=32)
    _conv_case(ops, 32, 64, (1, 4, 8, 8), (3, 3, 3), (1, 1, 1), gen)
    _conv_case(ops, 32, 32, (2, 7, 12, 7), (3, 3, 3), (1, 1, 1), gen, dy_scale=3e-7)
    _conv_case(ops, 32, 32, (1, 2, 16, 9), (3, 3, 3), (1, 1, 1), gen, dy_scale=2e4)


def test_pointwise_conv_weight_gradient_streaming_kernel(ops):
    """1x1x1 convs: dW comes from wgrad1x1_kernel (round 4): one, three (96 x 32), four (64 x 64) and nine (96 x 96: three tile groups)
    32 x 32 output tiles; row counts that are no multiple of a 128-row unit; more units than workgroups, so that a workgroup loops and
    the next unit's prefetch is exercised; gradients far below and above the fp16 range (this kernel stays on the f32-input MFMA: nothing
    is rescaled); a concatenated input (two launches into column slices of one dW)."""
    gen = torch.Generator().manual_seed(29)
    _conv_case(ops, 32, 96, (1, 5, 7, 3), (1, 1, 1), (1, 1, 1), gen)           # A = 96 (dy), Bc = 32: 105 rows
    _conv_case(ops, 64, 64, (2, 3, 11, 5), (1, 1, 1), (1, 1, 1), gen)          # four tiles in one workgroup
    _conv_case(ops, 96, 96, (1, 7, 3, 6), (1, 1, 1), (1, 1, 1), gen)           # nine tiles = three groups
    _conv_case(ops, 32, 32, (6, 45, 16, 9), (1, 1, 1), (1, 1, 1), gen)         # 77 760 rows = 608 units on 512 workgroups
    _conv_case(ops, 32, 32, (1, 9, 5, 3), (1, 1, 1), (1, 1, 1), gen, dy_scale=3e-7)
    _conv_case(ops, 32, 32, (1, 9, 5, 3), (1, 1, 1), (1, 1, 1), gen, dy_scale=2e4)
    _conv_case(ops, 96, 32, (2, 4, 6, 5), (1, 1, 1), (1, 1, 1), gen, split=64)  # skip concat: 64 + 32 input channels


def test_conv_transpose_backward(ops):
    gen = torch.Generator().manual_seed(22)
    for c, shp, kz, zs, op in ((32, (1, 4, 4, 2), 3, 2, (0, 0, 0)), (32, (2, 5, 4, 4), 3, 2, (0, 0, 1)),
                               (32, (1, 3, 3, 5), 4, 2, (0, 0, 0)), (64, (1, 4, 6, 3), 4, 2, (0, 0, 1)),
                               (32, (1, 2, 2, 7), 3, 2, (0, 1, 1)), (32, (2, 4, 6, 4), 3, 2, (0, 1, 0)),
                               (32, (1, 4, 3, 2), 3, 1, (0, 0, 0))):
        x = torch.randn((shp[0], c) + shp[1:], generator=gen, requires_grad=True)
        w = (torch.randn((c, c, kz, 4, 4), generator=gen) * 0.1).requires_grad_()
        b = torch.randn(c, generator=gen, requires_grad=True)
        y = O.cyl_conv_transpose3d(x, w, b, (zs, 2, 2), op)
        dy = torch.randn(y.shape, generator=gen)
        y.backward(dy)
        dx, dw, db = ops.conv_transpose_backward(cl(ops, x), w.detach().cuda(), cl(ops, dy), kz, zs, op)
        assert rel_l2(back(ops, dx), x.grad.numpy()) < TOL, ("dx", shp, kz, op)
        assert rel_l2(dw.cpu().numpy(), w.grad.numpy()) < TOL, ("dw", shp, kz, op)
        assert rel_l2(db.cpu().numpy(), b.grad.numpy()) < TOL, ("db", shp, kz, op)


def test_group_norm_backward(ops):
    gen = torch.Generator().manual_seed(23)
    for C, G, shp, silu in ((32, 8, (2, 5, 6, 4), True), (64, 8, (1, 23, 8, 4), True), (96, 8, (1, 3, 3, 5), True),
                            (64, 1, (2, 7, 3, 5), False), (32, 1, (1, 12, 4, 2), False),
                            # one-launch form beyond four row trips per thread (the second pass re-reads), 128 channels, 16 channels
                            (32, 8, (2, 10, 6, 5), True), (128, 8, (1, 5, 6, 4), True), (16, 4, (3, 4, 3, 5), True)):
        x = (torch.randn((shp[0], C) + shp[1:], generator=gen) * 2 + 0.7).requires_grad_()
        gm = torch.randn(C, generator=gen, requires_grad=True)
        bt = torch.randn(C, generator=gen, requires_grad=True)
        add = torch.randn(shp[0], C, generator=gen, requires_grad=True)
        z = F.group_norm(x, G, gm, bt, 1e-5)
        y = (F.silu(z) if silu else z) + add[:, :, None, None, None]
        dy = torch.randn(y.shape, generator=gen)
        y.backward(dy)
        dx, dg, db, dadd = ops.group_norm_backward(cl(ops, x), gm.detach().cuda(), bt.detach().cuda(), cl(ops, dy), G, silu=silu,
                                                   want_dadd=True)
        assert rel_l2(back(ops, dx), x.grad.numpy()) < TOL, ("dx", C, G)
        assert rel_l2(dg.cpu().numpy(), gm.grad.numpy()) < TOL, ("dgamma", C, G)
        assert rel_l2(db.cpu().numpy(), bt.grad.numpy()) < TOL, ("dbeta", C, G)
        assert rel_l2(dadd.cpu().numpy(), add.grad.numpy()) < TOL, ("dadd", C, G)


@pytest.mark.parametrize("nblk", [None, "3", "8"])
def test_conv_weight_gradient_z_sliding_kernel(ops, nblk, monkeypatch):
    """3x3x3 stride-1 dW from wgrad_ring_f16x2_kernel (round 4): a workgroup owns a chunk of consecutive units of one sample, x
    planes in a ring of 2 NZ + 2 slots, the next unit prefetched under the K loop.  Cases: one plane per unit (144-voxel planes:
    the ring wraps every four units), several planes per unit with a ragged last unit (23 planes of 32 voxels, NZ = 4), units
    whose 16-voxel K steps straddle planes (8-voxel planes), the normalised-x operand is covered by the training tests.
    CD_WGRAD_RING_NBLK caps the workgroups: long chunks (12+ units), more chunks than workgroups (a workgroup starts a second
    chunk with a used ring), and the default (one unit per chunk at these batch sizes: prologue only)."""
    if nblk:
        monkeypatch.setenv("CD_WGRAD_RING_NBLK", nblk)
    gen = torch.Generator().manual_seed(31)
    _conv_case(ops, 32, 32, (2, 45, 16, 9), (3, 3, 3), (1, 1, 1), gen)
    _conv_case(ops, 64, 64, (4, 23, 8, 4), (3, 3, 3), (1, 1, 1), gen)
    _conv_case(ops, 64, 32, (3, 12, 4, 2), (3, 3, 3), (1, 1, 1), gen)
    _conv_case(ops, 64, 32, (1, 7, 16, 9), (3, 3, 3), (1, 1, 1), gen, split=32)
    _conv_case(ops, 32, 32, (5, 9, 12, 7), (3, 3, 3), (1, 1, 1), gen, dy_scale=3e-7)
