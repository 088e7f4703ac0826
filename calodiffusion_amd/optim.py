"""Fused Adam for the training step: ``torch.optim.Adam`` as ``Train.train`` builds it (calodiffusion/train/train.py:144,
``Adam(model.parameters(), lr=LR)``; no amsgrad), all parameter tensors updated by ``cd_adam_step`` in ceil(n / 48) launches
instead of torch's per-operation foreach kernels.  State layout and ``state_dict`` keys are torch's (``step``, ``exp_avg``,
``exp_avg_sq``), so optimizer checkpoints interchange with ``torch.optim.Adam`` (``Train.pickup_checkpoint``, train.py:86-87)
and ``ReduceLROnPlateau`` (train.py:145-147) drives ``param_groups[i]['lr']`` as usual."""
import ctypes as C

import torch

from . import engine


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._lib = engine.load_library()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            steps = set()
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and p.grad.dtype == torch.float32):
                    raise RuntimeError("FusedAdam: parameters and gradients must be contiguous float32 device tensors")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.add(int(st["step"]))
            if len(steps) != 1:
                raise RuntimeError("FusedAdam: parameters of one group must share their step count")
            n = len(ps)
            arr = lambda xs: (C.c_void_p * n)(*[x.data_ptr() for x in xs])  # noqa: E731
            numel = (C.c_int64 * n)(*[p.numel() for p in ps])
            b1, b2 = group["betas"]
            engine._check(self._lib.cd_adam_step(n, arr(ps), arr([p.grad for p in ps]), arr([self.state[p]["exp_avg"] for p in ps]),
                                                 arr([self.state[p]["exp_avg_sq"] for p in ps]), numel, float(group["lr"]), float(b1),
                                                 float(b2), float(group["eps"]), float(group["weight_decay"]), steps.pop(),
                                                 engine._stream()))
            # cd_adam_step writes through raw pointers: tell torch (and UnetEngine.sync_weights, which keys its re-pack of the
            # plan's weight arena on the parameters' versions) that the tensors changed
            torch.autograd.graph.increment_version(ps)
        return loss
