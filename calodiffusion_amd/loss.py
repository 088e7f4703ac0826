"""Training objectives of the hot path (mirror of reference calodiffusion/models/loss.py for `hybrid_weight`)."""
from __future__ import annotations

import torch

from . import schedule
from .utils import allreduce_mean_, subsample_alphas


class Loss:
    """EDM scalings + noise-level draw (models/loss.py:9-142); the reductions of Loss._loss (l2 / l1 / mse / huber) run in
    cd_train_step / cd_loss_hybrid."""

    def __init__(self, config, n_steps, loss_type="l1") -> None:
        self.config = config
        self.update_step(n_steps)
        self.discrete_time = True
        self.P_mean, self.P_std, self.sigma_data = -1, 1, 0.5
        if "log" in config.get("NOISE_SCHED", "linear"):
            self.discrete_time = False
            self.P_mean, self.P_std, self.sigma_data = -1.2, 1.2, 1.0
        # Loss._loss (models/loss.py:97-116) raises at construction for an unknown LOSS_TYPE; the reductions themselves run on
        # the device (cd_train_step / cd_loss_hybrid)
        if loss_type not in self.LOSS_TYPES:
            raise NotImplementedError("Loss type %s not implemented, pick from (%s)" % (loss_type, self.LOSS_TYPES))
        self.loss_type = loss_type

    LOSS_TYPES = ("l1", "l2", "mse", "huber")

    def get_scaling(self, sigma):
        s2 = sigma ** 2 + self.sigma_data ** 2
        return {"c_skip": self.sigma_data ** 2 / s2, "c_out": sigma * self.sigma_data / s2 ** 0.5, "c_in": 1 / s2 ** 0.5}

    def update_step(self, steps: int):
        self.n_steps = steps
        tb = schedule.tables(steps)
        self.sqrt_alphas_cumprod = tb["sqrt_alphas_cumprod"]
        self.sqrt_one_minus_alphas_cumprod = tb["sqrt_one_minus_alphas_cumprod"]
        self.posterior_variance = tb["betas"] * (1.0 - tb["alphas_cumprod_prev"]) / (1.0 - tb["alphas_cumprod"])

    def draw_sigma(self, data, time=None, rnd_normal=None):
        """Noise level per sample, as Loss.__call__ draws it (models/loss.py:124-140)."""
        B = data.shape[0]
        if self.discrete_time:
            if time is None:
                time = torch.randint(0, self.n_steps, (B,), device=data.device).long()
            a = subsample_alphas(self.sqrt_alphas_cumprod, time, data.shape)
            b = subsample_alphas(self.sqrt_one_minus_alphas_cumprod, time, data.shape)
            return (b / a).reshape(B)
        if rnd_normal is None:
            rnd_normal = torch.randn((B,), device=data.device)
        return (rnd_normal * self.P_std + self.P_mean).exp().reshape(B)

    def __call__(self, model, data, E, noise=None, time=None, layers=None, rnd_normal=None):
        if noise is None:
            noise = torch.randn_like(data)
        sigma = self.draw_sigma(data, time=time, rnd_normal=rnd_normal)
        return self.loss_function(model, data, E, sigma=sigma, noise=noise, layers=layers)

    def loss_function(self, model, data, E, sigma=None, noise=None, layers=None):
        raise NotImplementedError

    def _device_loss(self, model, data, E, sigma, noise, layers):
        """What the three objective classes share: the engine's plan carries the objective (CdUnetDesc.objective, set from the class
        name in CaloDiffusion.init_model as the reference's denoise branches on it, calodiffusion.py:156-169), so one call
        evaluates pred / target / weight of that class and, in training, every parameter gradient."""
        cond = model.cond_tensor(E, layers)
        params = list(model.model.parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            # training: TrainDiffusion.training_loop calls loss.backward(); optimizer.step() on the result
            return _TrainStep.apply(model.engine(), self.loss_type, data, noise, sigma, cond, *params)
        return model.engine().loss_hybrid(data, noise, sigma, cond, self.loss_type)


class _TrainStep(torch.autograd.Function):
    """Loss value and parameter gradients from ONE call into the HIP library (cd_train_step); autograd only sees a
    node whose backward hands the pre-computed gradients (scaled by the incoming gradient) to the parameters."""

    @staticmethod
    def forward(ctx, engine, loss_type, data, noise, sigma, cond, *params):
        loss, flat = engine.train_step(data, noise, sigma, cond, loss_type)
        allreduce_mean_(flat)  # data parallel: identical replicas, one flat-buffer all-reduce per step
        ctx.engine, ctx.flat, ctx.params = engine, flat, params
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        # ONE scaling of the flat buffer (this call's own: train_step allocates it), then every parameter's .grad is a VIEW of it.
        # Handing the views back to autograd instead made AccumulateGrad clone each of the ~230 of them into a fresh .grad tensor --
        # one elementwise launch and one allocation per parameter and step (rocprofv3, round 3: 233 launches) -- so the hand-over
        # is done here, with torch's accumulation semantics: a parameter without a gradient takes the view, one that already
        # has a gradient (no zero_grad between two backward passes) gets the view added.
        flat = ctx.flat.mul_(grad_out.to(ctx.flat.dtype))
        for p, g in zip(ctx.params, ctx.engine.param_grads(flat)):
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = g
            else:
                p.grad = p.grad + g if p.grad.requires_grad else p.grad.add_(g)
        return (None,) * (6 + len(ctx.params))


class hybrid_weight(Loss):
    """x0-prediction with weight 1 + sigma^-2 (models/loss.py:163-179); value computed by cd_loss_hybrid, value and every
    gradient by cd_train_step."""

    def __init__(self, config, n_steps, loss_type="l1") -> None:
        super().__init__(config, n_steps, loss_type)

    def loss_function(self, model, data, E, sigma=None, noise=None, layers=None):
        return self._device_loss(model, data, E, sigma, noise, layers)


class noise_pred(Loss):
    """models/loss.py:181-196: the network predicts the noise; denoise returns x - sigma F (calodiffusion.py:161-162), the loss
    compares (data - (data - sigma denoise(x_noisy))) / sigma with the noise, unweighted."""

    def __init__(self, config, n_steps, loss_type="l1") -> None:
        super().__init__(config, n_steps, loss_type)

    def loss_function(self, model, data, E, sigma=None, noise=None, layers=None):
        return self._device_loss(model, data, E, sigma, noise, layers)


class mean_pred(Loss):
    """models/loss.py:198-210: the network output itself is the shower estimate (calodiffusion.py:164-165), weight sigma^-2."""

    def __init__(self, config, n_steps, loss_type="l1") -> None:
        super().__init__(config, n_steps, loss_type)

    def loss_function(self, model, data, E, sigma=None, noise=None, layers=None):
        return self._device_loss(model, data, E, sigma, noise, layers)


class minsnr(Loss):
    """models/loss.py:144-161, kept with the reference's signature: its __init__ takes no loss_type, so Diffusion.__init__
    (models/diffusion.py:30, which passes one) cannot construct it there either -- `TRAINING_OBJ: minsnr` raises the same
    TypeError in both -- and CaloDiffusion.denoise has no branch for it (`('hybrid' or 'minsnr') in name` tests 'hybrid' only)."""

    def __init__(self, config, n_steps) -> None:
        super().__init__(config, n_steps)

    def loss_function(self, model, data, E, sigma=None, noise=None, layers=None):
        raise ValueError("??? Training obj %s" % type(self).__name__)  # what model.denoise raises there (calodiffusion.py:169)
