"""Samplers of the hot path (mirror of reference calodiffusion/models/sample.py), resolved by class name from the config key
``SAMPLER`` exactly as the reference does (utils/utils.py:1047-1061).

The reference's loops call back into the model once or twice per step and do their elementwise updates (and, for DDim, five
host<->device ``extract`` round trips) in between.  Here every sampler is ONE C-ABI call:

* ``DDim`` / ``DDPM`` and the deterministic ``Euler`` ride ``cd_ddim_sample``: the per-step scalars are tabulated on the host
  once, one step is captured as a hipGraph and replayed;
* every other sampler is a *step program* for ``cd_sampler_run`` (include/calodiff.h): per step a short list of ops --
  linear combinations of a few (B,1,D,H,W) buffers, denoise calls, noise draws, trajectory records -- whose scalars are the
  columns of that step's row of a host table.  All of these samplers are exactly that: their updates are linear in
  {x, denoised, noise, history} with coefficients that depend on the step only.  Uniform programs replay one captured step
  graph; the others (Restart's nested loops, DPM-Solver-fast's changing orders) run their steps eagerly.

``DPMAdaptive`` decides every step on the host from a norm of the state: it is a host loop around ``denoise`` (one
``cd_denoise_safe`` call per model evaluation), not a step program.  ``DPMPPSDE`` / ``DPMPP2MSDE`` / ``DPMPP3MSDE`` are step
programs whose Brownian-tree noise (``torchsde`` in the reference) is drawn with the same law from the device Philox stream
(``_BrownianSDE``).  Not provided: ``BespokeNonStationary`` (needs a trained theta file); asking for it raises.
"""
from __future__ import annotations

import math
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import schedule
from .engine import SOP_DENOISE, SOP_LINCOMB, SOP_LINDIV, SOP_RANDN, SOP_RECORD


class Sample:
    def __init__(self, config) -> None:
        self.config = config
        self.sample_config = self.config.get("SAMPLER_OPTIONS", {})
        self.use_graph = bool(self.sample_config.get("HIP_GRAPH", True))
        self.seed = int(self.sample_config.get("SEED", 0))
        self.step_noise = None  # parity hook: the noise tensors to use, in draw order, instead of the device Philox stream
        self.noise_tensors_drawn = 0  # (B,1,D,H,W) tensors the last call took from the stream (Diffusion.sample advances by it)

    def __call__(self, model, start, energy, layers, num_steps, sample_offset, debug) -> Any:
        raise NotImplementedError

    @staticmethod
    def _stream(model, start):
        """(offset, stride) of the per-step noise in the model's Philox stream: right behind the start tensor (of the whole,
        possibly sharded, batch)."""
        if hasattr(model, "step_noise_stream"):
            return model.step_noise_stream(start)
        return getattr(model, "noise_offset", 0) + start.numel(), 0


class DDim(Sample):
    """Deterministic sampler (eta = 0), models/sample.py:29-109.  Returns (x, xs, x0s); the trajectories are only recorded
    when ``debug`` (the reference keeps 2N full tensors alive regardless and drops them in Diffusion.sample,
    diffusion.py:91,104)."""

    def __init__(self, config):
        super().__init__(config)
        self.ddim_eta = 0.0

    @torch.no_grad()
    def __call__(self, model, start, energy, layers, num_steps, sample_offset=0, debug=False) -> Any:
        table = schedule.ddim_step_table(num_steps, self.ddim_eta, sample_offset or 0)
        offset, stride = self._stream(model, start)
        x, xs, x0s = model.engine().ddim_sample(
            start, model.cond_tensor(energy, layers), table, step_noise=self.step_noise, seed=getattr(model, "noise_seed", self.seed),
            offset=offset, debug=debug, use_graph=self.use_graph, noise_stride=stride)
        self.noise_tensors_drawn = table.shape[0] if self.ddim_eta else 0
        if debug:
            return x, list(xs.unbind(0)), list(x0s.unbind(0))
        return x, [], []


class DDPM(DDim):
    """Stochastic version (eta = 1), models/sample.py:112-121."""

    def __init__(self, config):
        super().__init__(config)
        self.ddim_eta = 1.0


# ------------------------------------------------------------------------------------------------------------------
# step programs
# ------------------------------------------------------------------------------------------------------------------
X, NZ, XH, DN, X2, DN2, H0, H1, H2, H3 = range(10)  # buffer roles (0 = the running sample; 1 doubles as warm-up scratch)


class _Step:
    def __init__(self):
        self.ops: List[Tuple[int, int, Tuple[int, ...], int]] = []
        self.coefs: List[float] = []

    def lin(self, dst: int, terms: Sequence[Tuple[int, float]]):
        """buf[dst] = sum coef * buf[src]"""
        assert 1 <= len(terms) <= 6
        self.ops.append((SOP_LINCOMB, dst, tuple(b for b, _ in terms), len(self.coefs)))
        self.coefs += [float(c) for _, c in terms]

    def lin_div(self, dst: int, terms: Sequence[Tuple[int, float]], div: float = 1.0):
        """buf[dst] = ((c0 * buf[s0] + c1 * buf[s1]) + ...) / div in the operation order of a chain of torch elementwise ops: every
        product, every sum (left to right) and the division rounded to fp32 on its own, nothing fused."""
        assert 1 <= len(terms) <= 6
        self.ops.append((SOP_LINDIV, dst, tuple(b for b, _ in terms), len(self.coefs)))
        self.coefs += [float(c) for _, c in terms] + [float(div)]

    def denoise(self, dst: int, src: int, sigma: float):
        self.ops.append((SOP_DENOISE, dst, (src,), len(self.coefs)))
        self.coefs.append(float(sigma))

    def randn(self, dst: int):
        self.ops.append((SOP_RANDN, dst, (), 0))

    def record(self, which: int, src: int):
        """which: 0 = xs, 1 = x0s"""
        self.ops.append((SOP_RECORD, which, (src,), 0))


class Program:
    """A sampler as data for cd_sampler_run: buffers, per-step op lists and the host coefficient table."""

    def __init__(self, n_bufs: int, start_scale: float):
        self.n_bufs, self.start_scale = int(n_bufs), float(start_scale)
        self._steps: List[_Step] = []
        self.ops = self.op_begin = self.coefs = None
        self.n_randn = 0

    def step(self) -> _Step:
        self._steps.append(_Step())
        return self._steps[-1]

    def finalize(self) -> "Program":
        assert self._steps, "empty sampler program"
        width = max(1, max(len(s.coefs) for s in self._steps))
        with np.errstate(over="ignore"):  # (non-finite coefficients are legitimate: the reference divides by t_next = 0 too)
            self.coefs = np.array([s.coefs + [0.0] * (width - len(s.coefs)) for s in self._steps], dtype=np.float64).astype(np.float32)
        self.n_randn = sum(1 for s in self._steps for o in s.ops if o[0] == SOP_RANDN)
        if all(s.ops == self._steps[0].ops for s in self._steps):
            self.ops, self.op_begin = list(self._steps[0].ops), None
        else:
            self.ops, self.op_begin = [], [0]
            for s in self._steps:
                self.ops += s.ops
                self.op_begin.append(len(self.ops))
        return self


class _ProgramSampler(Sample):
    """Shared call path of the step-program samplers."""

    returns_trajectories = True

    def build(self, model, num_steps: int, sample_offset: int) -> Program:
        raise NotImplementedError

    @torch.no_grad()
    def __call__(self, model, start, energy, layers, num_steps, sample_offset=0, debug=False) -> Any:
        eng = model.engine()
        if not hasattr(eng, "sampler_run"):
            raise NotImplementedError(f"{type(self).__name__} runs on the U-Net engine; the layer model's one-launch sampler "
                                      "offers DDim / DDPM / Euler")
        prog = self.build(model, num_steps, sample_offset or 0).finalize()
        offset, stride = self._stream(model, start)
        x, xs, x0s = eng.sampler_run(start, model.cond_tensor(energy, layers), prog, step_noise=self.step_noise,
                                     seed=getattr(model, "noise_seed", self.seed), offset=offset, noise_stride=stride,
                                     debug=debug and self.returns_trajectories, use_graph=self.use_graph)
        self.noise_tensors_drawn = prog.n_randn
        return self.finish(x, xs, x0s, debug)

    def finish(self, x, xs, x0s, debug):
        if debug and self.returns_trajectories:
            return x, [] if xs is None else list(xs.unbind(0)), [] if x0s is None else list(x0s.unbind(0))
        return x, [], []


def _f(v) -> float:
    return float(v)


class EDMAbstract(_ProgramSampler):
    """EDM samplers on the Karras noise schedule (models/sample.py:577-727): options and the common step prologue
    `t_hat = t_cur (1 + gamma)`, `x_hat = x_cur + sqrt(t_hat^2 - t_cur^2) S_noise randn`, `denoised = D(x_hat, t_hat)`."""

    def __init__(self, config) -> None:
        super().__init__(config)
        noisy = self.config.get("NOISY_SAMPLE", False)
        self.S_churn = 40 if noisy else 0
        self.S_min = self.sample_config.get("S_MIN", 0.01)
        self.S_max = 50 if noisy else 1
        self.S_noise = self.sample_config.get("S_NOISE", 1.003)
        self.sigma_min = self.sample_config.get("SIGMA_MIN", 0.002)
        self.sigma_max = self.sample_config.get("SIGMA_MAX", 80.0)
        self.orig_schedule = self.sample_config.get("ORG_SCHEDULE", False)
        self.rho = self.sample_config.get("RHO", 7)
        self.order = self.sample_config.get("ORDER", 4)
        self.restart_gamma = self.sample_config.get("RESTART_GAMMA", 0.05)
        if self.orig_schedule:
            raise NotImplementedError("ORG_SCHEDULE (iDDPM time steps) is not provided: the reference's own branch calls "
                                      "alpha_bar() with a missing argument (models/sample.py:664,693)")

    def t_steps(self, num_steps, sample_offset) -> torch.Tensor:
        return schedule.edm_time_steps(num_steps, sample_offset, sigma_min=self.sigma_min, sigma_max=self.sigma_max, rho=self.rho)

    def churn(self, t_cur: torch.Tensor, num_steps: int, gamma_on: Optional[float] = None):
        """(t_hat, noise coefficient) of the 'increase noise temporarily' prologue (models/sample.py:642-651), fp32 like the
        reference's 0-dim tensor arithmetic."""
        g = min(self.S_churn / num_steps, np.sqrt(2) - 1) if gamma_on is None else gamma_on
        gamma = g if self.S_min <= t_cur <= self.S_max else 0
        t_hat = torch.as_tensor(t_cur + gamma * t_cur)
        return t_hat, (t_hat ** 2 - t_cur ** 2).sqrt() * self.S_noise

    def prologue(self, st: _Step, t_cur, num_steps, noisy_prog: bool, record=True):
        """Ops of the common prologue; returns (t_hat, buffer holding x_hat)."""
        t_hat, cn = self.churn(t_cur, num_steps)
        if record:
            st.record(0, X)  # xs.append(x_cur)
        xh = X
        if noisy_prog:  # one draw per step, as the reference (its randn_like is unconditional)
            st.randn(NZ)
            st.lin(XH, [(X, 1.0), (NZ, _f(cn))])
            xh = XH
        st.denoise(DN, xh, _f(t_hat))
        if record:
            st.record(1, DN)  # x0s.append(denoised)
        return t_hat, xh

    def in_loop(self, st: _Step, xh: int, t_hat, t_next):
        raise NotImplementedError

    def build(self, model, num_steps, sample_offset) -> Program:
        t = self.t_steps(num_steps, sample_offset)
        prog = Program(6, _f(t[0]))
        noisy = self.S_churn > 0
        for t_cur, t_next in zip(t[:-1], t[1:]):
            st = prog.step()
            t_hat, xh = self.prologue(st, t_cur, num_steps, noisy)
            self.in_loop(st, xh, t_hat, t_next)
        return prog


def _euler_terms(xh: int, t_hat, t_to) -> List[Tuple[int, float]]:
    """x_hat + (t_to - t_hat) (x_hat - D) / t_hat  as  a x_hat + b D"""
    r = _f(t_to) / _f(t_hat)
    return [(xh, r), (DN, 1.0 - r)]


class Euler(EDMAbstract):
    """EDM first-order sampler (models/sample.py:771-789).  Without NOISY_SAMPLE the update x0 + t_next (x - x0)/t is the
    device loop of DDim with the step table (t_i, t_{i+1}, 0, 1) and rides cd_ddim_sample; with it (S_churn = 40: noise is
    added back before every denoise call) it is a step program."""

    @torch.no_grad()
    def __call__(self, model, start, energy, layers, num_steps, sample_offset=0, debug=False) -> Any:
        if self.S_churn > 0:
            return super().__call__(model, start, energy, layers, num_steps, sample_offset, debug)
        table = schedule.edm_euler_step_table(num_steps, sample_offset or 0, sigma_min=self.sigma_min, sigma_max=self.sigma_max,
                                              rho=self.rho)
        x, xs, x0s = model.engine().ddim_sample(start, model.cond_tensor(energy, layers), table, debug=debug,
                                                use_graph=self.use_graph)
        self.noise_tensors_drawn = 0
        if debug:
            return x, list(xs.unbind(0)), list(x0s.unbind(0))
        return x, [], []

    def in_loop(self, st, xh, t_hat, t_next):
        st.lin(X, _euler_terms(xh, t_hat, t_next))


class Heun(EDMAbstract):
    """EDM second-order sampler as the reference computes it (models/sample.py:792-822): Euler predictor to t_next, a second
    denoise there, and the average of the two slopes -- where the second slope is taken from `self.x_next`, which at that
    point still holds x_cur (the state before the churn), not the predictor: d' = (x_cur - D(x', t_next)) / t_next.
    Mirrored as is.  Like the reference, the last step (t_next = 0) divides by zero: the final tensor is not finite there;
    use the trajectories (debug) or Euler/LMS for a finite end point."""

    def in_loop(self, st, xh, t_hat, t_next):
        st.lin(X2, _euler_terms(xh, t_hat, t_next))      # x' = x_hat + h d_cur
        st.denoise(DN2, X2, _f(t_next))
        th, tn = _f(t_hat), _f(t_next)
        h = tn - th
        with np.errstate(divide="ignore", invalid="ignore"):
            c = np.float64(0.5 * h) / np.float64(tn)
        # x_hat + h (0.5 (x_hat - D)/t_hat + 0.5 (x_cur - D')/t_next)
        st.lin(X, [(xh, 1.0 + 0.5 * h / th), (DN, -0.5 * h / th), (X, float(c)), (DN2, float(-c))])


class DPM2(EDMAbstract):
    """DPM-Solver-2 in the EDM loop (models/sample.py:824-851): midpoint in log-sigma.  t_next = 0 on the last step makes
    t_mid = 0 and the final tensor non-finite, as in the reference."""

    def in_loop(self, st, xh, t_hat, t_next):
        t_mid = t_hat.log().lerp(t_next.log(), 0.5).exp()
        th, tm, tn = _f(t_hat), _f(t_mid), _f(t_next)
        st.lin(X2, _euler_terms(xh, t_hat, t_mid))       # x_2 = x_hat + d_cur (t_mid - t_hat)
        st.denoise(DN2, X2, tm)
        with np.errstate(divide="ignore", invalid="ignore"):
            c = np.float64(tn - th) / np.float64(tm)
        st.lin(X, [(xh, 1.0), (X2, float(c)), (DN2, float(-c))])  # x_hat + h (x_2 - D_2) / t_mid


def _lms_coeff(order: int, t: np.ndarray, i: int, j: int) -> float:
    """Integral over [t_i, t_{i+1}] of the j-th Lagrange basis polynomial on the nodes t_i, t_{i-1}, ... (the reference
    integrates the same polynomial numerically, utils/sampling.py:77-92; it is done in closed form here)."""
    poly = np.poly1d([1.0])
    for k in range(order):
        if k != j:
            poly = poly * np.poly1d([1.0, -t[i - k]]) / (t[i - j] - t[i - k])
    anti = poly.integ()
    return float(anti(t[i + 1]) - anti(t[i]))


class LMS(EDMAbstract):
    """Linear multistep sampler (models/sample.py:729-769): x_next = x_hat + sum_j c_j d_{i-j} over up to ORDER stored slopes
    d = (x - D)/t.  No churn and no trajectories, as in the reference (its xs / x0s lists stay empty)."""

    returns_trajectories = False

    def build(self, model, num_steps, sample_offset) -> Program:
        t = self.t_steps(num_steps, sample_offset)
        tn = t.double().numpy()
        order = int(self.order)
        if not 1 <= order <= 4:
            raise ValueError("LMS: ORDER must be 1..4")
        hist = [H0, H1, H2, H3][:order]
        prog = Program(10, _f(t[0]))
        for i in range(len(t) - 1):
            st = prog.step()
            st.denoise(DN, X, _f(t[i]))
            for k in range(order - 1, 0, -1):  # shift the history: d_{i-k} <- d_{i-k+1}
                st.lin(hist[k], [(hist[k - 1], 1.0)])
            st.lin(H0, [(X, 1.0 / _f(t[i])), (DN, -1.0 / _f(t[i]))])
            cur = min(i + 1, order)
            cs = [_lms_coeff(cur, tn, i, j) if j < cur else 0.0 for j in range(order)]
            st.lin(X, [(X, 1.0)] + [(hist[j], cs[j]) for j in range(order)])
        return prog


class Restart(EDMAbstract):
    """Restart sampler (models/sample.py:853-954, arXiv 2306.14878): the churned Euler main loop, and after main step i a
    number of 'restart' excursions when i + 1 is a key of RESTART_LIST = {i: [N_restart, K, t_min, t_max]}: noise is added back
    up to t_max and a Heun sub-trajectory of N_restart Karras steps comes back down to t_{i+1}.

    Key types matter exactly as in the reference: it tests `index + 1 in restart_list.keys()` with an int, so the default
    table -- whose keys are strings -- never triggers and the sampler is the (churned) Euler loop; integer keys (a YAML/JSON
    config with unquoted keys) switch the excursions on.  The program is non-uniform, its steps run eagerly."""

    def __init__(self, config):
        super().__init__(config)
        default_restart = {"0": [4, 1, 19.35, 40.79], "1": [4, 1, 1.09, 1.92], "2": [4, 4, 0.59, 1.09],
                           "3": [4, 1, 0.30, 0.59], "4": [4, 4, 0.06, 0.30]}
        self.restart_list = self.sample_config.get("RESTART_LIST", default_restart)

    def build(self, model, num_steps, sample_offset) -> Program:
        t = self.t_steps(num_steps, sample_offset)
        prog = Program(6, _f(t[0]))
        self._main_steps = []
        for index, (t_cur, t_next) in enumerate(zip(t[:-1], t[1:])):
            self._main_steps.append(len(prog._steps))
            st = prog.step()
            # main step: the reference draws the churn noise unconditionally (coefficient 0 without NOISY_SAMPLE)
            t_hat, cn = self.churn(t_cur, num_steps)
            st.randn(NZ)
            st.lin(XH, [(X, 1.0), (NZ, _f(cn))])
            st.denoise(DN, XH, _f(t_hat))
            st.record(1, DN)
            st.lin(X, _euler_terms(XH, t_hat, t_next))
            st.record(0, X)
            if index + 1 not in self.restart_list.keys():
                continue
            n_restart, k_rep, _t_min, t_max = self.restart_list[index + 1]
            for _ in range(int(k_rep)):
                new_t = schedule.karras_steps(int(n_restart), t[index + 1], t_max, self.rho)
                total = len(new_t)
                st = prog.step()
                st.randn(NZ)  # x_next += randn sqrt(t_0^2 - t_last^2) S_noise
                st.lin(X, [(X, 1.0), (NZ, _f((new_t[0] ** 2 - new_t[-1] ** 2).sqrt() * self.S_noise))])
                for j, (tc, tn) in enumerate(zip(new_t[:-1], new_t[1:])):
                    if j:
                        st = prog.step()
                    th, cn = self.churn(tc, num_steps, gamma_on=self.restart_gamma)
                    st.randn(NZ)
                    st.lin(XH, [(X, 1.0), (NZ, _f(cn))])
                    st.denoise(DN, XH, _f(th))
                    if j < total - 2 or new_t[-1] != 0:  # second-order correction
                        st.lin(X2, _euler_terms(XH, th, tn))
                        st.denoise(DN2, X2, _f(tn))
                        h = _f(tn) - _f(th)
                        # x_hat + h (0.5 (x_hat - D)/t_hat + 0.5 (x' - D')/t_next)
                        st.lin(X, [(XH, 1.0 + 0.5 * h / _f(th)), (DN, -0.5 * h / _f(th)), (X2, 0.5 * h / _f(tn)),
                                   (DN2, -0.5 * h / _f(tn))])
                    else:
                        st.lin(X, _euler_terms(XH, th, tn))
        return prog

    def finish(self, x, xs, x0s, debug):
        # (the reference appends the builtin `next` to xs, sample.py:951: its xs is unusable; here xs = x_next of every main
        # step).  Only main steps record; their slots are picked out of the per-program-step trajectories.
        if debug and xs is not None:
            xs, x0s = xs[self._main_steps], x0s[self._main_steps]
        return super().finish(x, xs, x0s, debug)


# ------------------------------------------------------------------------------------------------------------------
# DPM-Solver family on the model's own (cosine-schedule) noise levels, models/sample.py:124-186
# ------------------------------------------------------------------------------------------------------------------
class DPM(_ProgramSampler):
    """DPM-Solver-fast with a fixed step size (models/sample.py:124-186, utils/sampling.py:385-506): num_steps function
    evaluations spread over floor(nfe/3)+1 intervals of orders 3,...,3,2,1 (or 3,...,3,nfe%3), uniform in t = -log sigma
    between the largest and the smallest noise level of the model's schedule."""

    returns_trajectories = False

    def __init__(self, config):
        super().__init__(config)
        self.eta = self.sample_config.get("ETA", 0)
        self.s_noise = self.sample_config.get("S_NOISE", 1.0)

    @staticmethod
    def sigma_fn(t):
        return t.neg().exp()

    @staticmethod
    def time_fn(t):
        return t.log().neg()

    def create_sigmas(self, model, num_steps) -> torch.Tensor:
        lf = model.loss_function
        return torch.tensor([lf.sqrt_one_minus_alphas_cumprod[num_steps - t - 1] / lf.sqrt_alphas_cumprod[num_steps - t - 1]
                             for t in torch.arange(num_steps)])

    def setup_sigmas(self, model, num_steps) -> torch.Tensor:
        """DPM.setup (sample.py:155-162), including its side effect on the model's schedule tables."""
        if model.nsteps != num_steps:
            model.loss_function.update_step(num_steps)
        return self.create_sigmas(model, num_steps)

    def build(self, model, num_steps, sample_offset) -> Program:
        sig = self.setup_sigmas(model, num_steps)
        sigma_min, sigma_max = sig[-1], sig[0]
        if sigma_min <= 0 or sigma_max <= 0:
            raise ValueError("sigma_min and sigma_max must not be 0")
        if self.eta:
            raise NotImplementedError("DPM: ETA > 0 (ancestral noise) is not provided")
        t_start, t_end = self.time_fn(sigma_max.clone()), self.time_fn(sigma_min.clone())
        nfe = num_steps
        m = math.floor(nfe / 3) + 1
        ts = torch.linspace(t_start, t_end, m + 1)
        orders = [3] * (m - 2) + [2, 1] if nfe % 3 == 0 else [3] * (m - 1) + [nfe % 3]
        prog = Program(10, _f(sig[0]))
        EPS, U1, EPS1, U2, EPS2, DIF = H0, X2, H1, H2, H3, XH
        sg = self.sigma_fn

        # Every update below is emitted in the reference's own operation order (DPMSolver.eps / dpm_solver_{1,2,3}_step,
        # utils/sampling.py:402-456) as LINDIV ops -- products, sums and the division each rounded to fp32, scalars combined in
        # fp32 torch arithmetic exactly as the reference's 0-dim tensors are: the higher-order steps subtract terms of order
        # sigma_max from each other and amplify the rounding of u1 / u2 by sigma(t) / sigma(s1), so a re-associated update
        # (x / sigma - D / sigma, folded coefficients) that is just as accurate still ends 1.5e-4 away from the reference.
        def eps_of(st, dst, src, tt):  # eps = (x - model(x, sigma(t))) / sigma(t)                                  :402-411
            st.denoise(DN, src, _f(sg(tt)))
            st.lin_div(dst, [(src, 1.0), (DN, -1.0)], _f(sg(tt)))

        for i, order in enumerate(orders):
            t, t_next = ts[i], ts[i + 1]
            h = t_next - t
            st = prog.step()
            eps_of(st, EPS, X, t)
            a = sg(t_next) * h.expm1()  # x - sigma(t_next) * h.expm1() * eps: the scalar product first, as python evaluates it
            if order == 1:                                                                                        # :413-418
                st.lin_div(X, [(X, 1.0), (EPS, -_f(a))])
                continue
            r1 = 1 / 2 if order == 2 else 1 / 3
            s1 = t + r1 * h
            st.lin_div(U1, [(X, 1.0), (EPS, -_f(sg(s1) * (r1 * h).expm1()))])
            eps_of(st, EPS1, U1, s1)
            st.lin_div(DIF, [(EPS1, 1.0), (EPS, -1.0)])  # (eps_r1 - eps)
            if order == 2:                                                                                        # :420-432
                b = sg(t_next) / (2 * r1) * h.expm1()
                st.lin_div(X, [(X, 1.0), (EPS, -_f(a)), (DIF, -_f(b))])
                continue
            r2 = 2 / 3                                                                                            # :434-456
            s2 = t + r2 * h
            c = sg(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1)
            st.lin_div(U2, [(X, 1.0), (EPS, -_f(sg(s2) * (r2 * h).expm1())), (DIF, -_f(c))])
            eps_of(st, EPS2, U2, s2)
            st.lin_div(DIF, [(EPS2, 1.0), (EPS, -1.0)])  # (eps_r2 - eps)
            b = sg(t_next) / r2 * (h.expm1() / h - 1)
            st.lin_div(X, [(X, 1.0), (EPS, -_f(a)), (DIF, -_f(b))])
        return prog

    def finish(self, x, xs, x0s, debug):
        return x, None, None  # sample.py:185


class DPMPP2S(DPM):
    """DPM-Solver++(2S) (models/sample.py:311-344): two denoise calls per step.  As in the reference the ancestral noise is
    added ONCE, after the loop, with the last step's sigma_up (its `if` is outside the loop body)."""

    def build(self, model, num_steps, sample_offset) -> Program:
        sig = self.setup_sigmas(model, num_steps)
        prog = Program(6, _f(sig[0]))
        sigma_up = 0.0
        for i in range(len(sig) - 1):
            if not self.eta:
                sigma_down, sigma_up = sig[i + 1], 0.0
            else:
                sigma_up = min(sig[i + 1], self.eta * (sig[i + 1] ** 2 * (sig[i] ** 2 - sig[i + 1] ** 2) / sig[i] ** 2) ** 0.5)
                sigma_down = (sig[i + 1] ** 2 - sigma_up ** 2) ** 0.5
            t, t_next = self.time_fn(sig[i]), self.time_fn(torch.as_tensor(sigma_down))
            r = 1 / 2
            h = t_next - t
            sm = t + r * h
            st = prog.step()
            st.denoise(DN, X, _f(sig[i]))
            st.lin(X2, [(X, _f(self.sigma_fn(sm) / self.sigma_fn(t))), (DN, -_f((-h * r).expm1()))])
            st.denoise(DN2, X2, _f(self.sigma_fn(sm)))
            st.lin(X, [(X, _f(self.sigma_fn(t_next) / self.sigma_fn(t))), (DN2, -_f((-h).expm1()))])
        if len(sig) > 1 and sig[-1] > 0 and _f(sigma_up) != 0.0:
            st.randn(NZ)
            st.lin(X, [(X, 1.0), (NZ, _f(self.s_noise * sigma_up))])
        return prog


class DPMPP2M(DPM):
    """DPM-Solver++(2M) (models/sample.py:415-449): one denoise call per step and the previous step's denoised."""

    def build(self, model, num_steps, sample_offset) -> Program:
        sig = self.setup_sigmas(model, num_steps)
        prog = Program(6, _f(sig[0]))
        OLD = X2
        for i in range(len(sig) - 1):
            t, t_next = self.time_fn(sig[i]), self.time_fn(sig[i + 1])
            h = t_next - t
            a, b = _f(self.sigma_fn(t_next) / self.sigma_fn(t)), -_f((-h).expm1())
            st = prog.step()
            st.denoise(DN, X, _f(sig[i]))
            if i == 0 or sig[i + 1] == 0:
                st.lin(X, [(X, a), (DN, b), (OLD, 0.0)])
            else:
                r = _f((t - self.time_fn(sig[i - 1])) / h)
                st.lin(X, [(X, a), (DN, b * (1 + 1 / (2 * r))), (OLD, -b / (2 * r))])
            st.lin(OLD, [(DN, 1.0)])
        return prog


class Consistency(_ProgramSampler):
    """Multistep consistency sampling (models/sample.py:957-1011, utils/sampling.py:1143-1173): denoise, re-noise to the next
    (hard-coded) level, repeat.  Returns (x, xs, x0) with x0 the LAST denoised tensor, as the reference does."""

    def __init__(self, config) -> None:
        super().__init__(config)
        self.consis_nsteps = self.config.get("CONSIS_NSTEPS", 100)

    def build(self, model, num_steps, sample_offset) -> Program:
        n = self.consis_nsteps
        orig = model.nsteps
        lf = model.loss_function
        lf.update_step(n)
        idx = [0, int(round(n * 0.5)), int(round(n * 0.7)), int(round(n * 0.9)), int(round(n * 0.95))]
        t_all = [lf.sqrt_one_minus_alphas_cumprod[n - t - 1] / lf.sqrt_alphas_cumprod[n - t - 1] for t in range(n)]
        t_steps = torch.tensor([t_all[i] for i in idx[:num_steps]]) if num_steps > 1 else torch.tensor([t_all[0]])
        sig = torch.cat([t_steps, torch.zeros_like(t_steps[:1])])
        lf.update_step(orig)
        sigma_min = 0.002
        prog = Program(6, _f(sig[0]))
        for s_cur, s_next in zip(sig[:-1], sig[1:]):
            st = prog.step()
            st.denoise(DN, X, _f(s_cur))
            s_next = torch.clip(s_next, sigma_min, None)
            if s_next > sigma_min:
                st.randn(NZ)
                st.lin(X, [(DN, 1.0), (NZ, _f(torch.sqrt(s_next ** 2 - sigma_min ** 2)))])
            else:
                st.lin(X, [(DN, 1.0)])
            st.record(1, DN)
            st.record(0, X)
        return prog

    def finish(self, x, xs, x0s, debug):
        if debug:
            return x, [] if xs is None else list(xs.unbind(0)), None if x0s is None else x0s[-1]
        return x, [], None


def _unavailable(name, why):
    class _Missing(Sample):
        def __init__(self, config):
            raise NotImplementedError(f"sampler {name!r} is not provided on the HIP path: {why}")
    _Missing.__name__ = name
    return _Missing


class DPMAdaptive(DPM):
    """DPM-Solver-12 / -23 with an error test per step (models/sample.py:188-309) as a host loop around the denoiser: every step
    the host compares the lower- and the higher-order estimate and decides, so it cannot be a device step program.

    The reference's own class cannot run: it hands the denoiser a (B,) sigma, which `x * c_in` only broadcasts for B = 1
    (calodiffusion.py:159), and it unpacks `torch.randn_like(x)` into `lambda_0, lambda_s` (sample.py:255, 305), which needs
    B = 2 -- every call raises.  There is therefore no reference trajectory to pin this class to; it implements what the
    reference's code would compute with those two defects removed (the lambdas never influence the result), including its other
    quirks:

    * `PIDStepSizeControl.update_h` returns the new step size and the caller drops it (utils/sampling.py:1281-1290): the step
      stays H_INIT for the whole trajectory, the controller only accepts or rejects;
    * a rejected step would therefore repeat forever with the same step size: RuntimeError here;
    * with ETA = 0 the ancestral noise has amplitude sqrt(sigma_t^2 - sigma_t'^2) = 0: the trajectory is deterministic.

    Three denoiser calls per step at order 3 (two at order 2), each one cd_denoise_safe call.  Pinned by the same loop on the
    CPU oracle (tests/test_host.py, tests/test_gpu_round3.py)."""

    def __init__(self, config):
        super().__init__(config)
        sc = self.sample_config
        self.order = sc.get("ORDER", 3)
        self.r_tol, self.a_tol = sc.get("R_TOL", 0.05), sc.get("A_TOL", 0.0078)
        self.h_init, self.t_err = sc.get("H_INIT", 0.05), sc.get("T_ERROR", 1e-5)
        self.accept_safety = sc.get("ACCEPT_SAFETY", 0.81)

    def build(self, model, num_steps, sample_offset):
        raise NotImplementedError("DPMAdaptive decides its steps on the host: it has no device step program")

    def _eps(self, model, x, t, energy, layers):
        sigma = self.sigma_fn(t)
        den = model.denoise(x, E=energy, sigma=(sigma * x.new_ones([x.shape[0]])), layers=layers)
        self.denoise_calls += 1
        return (x - den) / sigma

    @torch.no_grad()
    def __call__(self, model, start, energy, layers, num_steps, sample_offset=0, debug=False):
        sig = self.setup_sigmas(model, num_steps)
        x = start * sig[0]
        sigma_min, sigma_max = sig[-1], sig[0]
        if sigma_min <= 0 or sigma_max <= 0:
            raise ValueError("sigma_min and sigma_max must not be 0")
        if self.order not in {2, 3}:
            raise ValueError("order should be 2 or 3")
        if self.eta:
            raise NotImplementedError("DPMAdaptive: ETA > 0 (ancestral noise) is not provided")
        t_start, t_end = self.time_fn(sigma_max.clone()), self.time_fn(sigma_min.clone())
        forward = bool(t_end > t_start)
        h = torch.tensor(abs(self.h_init) * (1 if forward else -1))
        atol, rtol = torch.tensor(self.a_tol, device=x.device), torch.tensor(self.r_tol, device=x.device)
        s_fn = self.sigma_fn
        self.denoise_calls = self.steps_taken = 0
        s, x_prev = t_start, x
        while (s < t_end - self.t_err) if forward else (s > self.t_err + self.t_err):
            t = torch.minimum(t_end, s + h) if forward else torch.maximum(t_end, s + h)
            tp = torch.minimum(t_end, t)  # eta = 0: sigma_down = sigma(t)
            hh = tp - s
            eps = self._eps(model, x, s, energy, layers)
            if self.order == 2:
                x_low = x - _f(s_fn(tp) * hh.expm1()) * eps
                r1 = 1 / 2
            else:
                r1 = 1 / 3
            s1 = s + r1 * hh
            u1 = x - _f(s_fn(s1) * (r1 * hh).expm1()) * eps
            eps_r1 = self._eps(model, u1, s1, energy, layers)
            x_2 = x - _f(s_fn(tp) * hh.expm1()) * eps - _f(s_fn(tp) / (2 * r1) * hh.expm1()) * (eps_r1 - eps)
            if self.order == 2:
                x_high = x_2
            else:
                x_low, r2 = x_2, 2 / 3
                s2 = s + r2 * hh
                u2 = x - _f(s_fn(s2) * (r2 * hh).expm1()) * eps - _f(s_fn(s2) * (r2 / r1) * ((r2 * hh).expm1() / (r2 * hh) - 1)) * (eps_r1 - eps)
                eps_r2 = self._eps(model, u2, s2, energy, layers)
                x_high = x - _f(s_fn(tp) * hh.expm1()) * eps - _f(s_fn(tp) / r2 * (hh.expm1() / hh - 1)) * (eps_r2 - eps)
            delta = torch.maximum(atol, rtol * torch.maximum(x_low.abs(), x_prev.abs()))
            error = torch.linalg.norm((x_low - x_high) / delta) / x.numel() ** 0.5
            if not bool(torch.all(error <= 1.0)):
                raise RuntimeError(f"DPMAdaptive: the step at t = {float(s):.4f} (sigma {float(s_fn(s)):.4g}) is rejected (error "
                                   f"{float(error):.3f} > 1) and the reference's controller never changes its step size "
                                   f"(PIDStepSizeControl.update_h's result is dropped): the reference loops forever here; "
                                   f"lower H_INIT or raise R_TOL / A_TOL")
            x_prev, x, s = x_low, x_high, t
            self.steps_taken += 1
        return x, None, None


def _ancestral_step(sigma_from, sigma_to, eta):
    """get_ancestral_step (utils/sampling.py:31-41): (sigma_down, sigma_up)."""
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    return (sigma_to ** 2 - sigma_up ** 2) ** 0.5, sigma_up


class _BrownianSDE(DPM):
    """The stochastic DPM-Solver++ samplers (models/sample.py:347-574).  The reference draws their noise from
    `sampling.BrownianTreeNoiseSampler` (utils/sampling.py:356-382, a torchsde Brownian tree over sigma): a call
    noise(sigma, sigma') returns (W(sigma') - W(sigma)) / sqrt|sigma' - sigma| -- a unit normal tensor, and draws over disjoint sigma
    intervals are independent.  torchsde is not installed here and nothing in the tree fixes the VALUES of a path, only its law,
    so the step programs build the same law from the device Philox stream: one fresh unit normal per disjoint interval, in the
    order the sampler walks down the schedule, and -- where a sampler asks for two overlapping intervals in one step (DPMPPSDE:
    [sigma_s, sigma_i] and [sigma_next, sigma_i]) -- the second draw as the variance-weighted sum the Brownian path implies,
        n(sigma_i -> sigma_next) = (sqrt(d1) n(sigma_i -> sigma_s) + sqrt(d2) n(sigma_s -> sigma_next)) / sqrt(d1 + d2).
    With ETA = 0 (the DPM family's default) every noise coefficient is exactly zero: the reference still calls its sampler and
    multiplies by 0.0, the programs draw nothing (and then replay one captured step graph).  There is no reference trajectory to pin
    these classes to (the reference cannot construct them here): they are pinned by the same loops restated on the CPU oracle with
    the same unit normals (oracle/samplers_oracle.py: dpmpp_sde / dpmpp_2m_sde / dpmpp_3m_sde)."""


class DPMPPSDE(_BrownianSDE):
    """DPM-Solver++ (stochastic), models/sample.py:347-416: two denoise calls per step, ancestral split of each half step.
    SAMPLER_OPTIONS: R (0.5), ETA (0), S_NOISE (1)."""

    def __init__(self, config):
        super().__init__(config)
        self.r = self.sample_config.get("R", 0.5)

    def build(self, model, num_steps, sample_offset) -> Program:
        sig = self.setup_sigmas(model, num_steps)
        prog = Program(6, _f(sig[0]))
        sg, tf = self.sigma_fn, self.time_fn
        for i in range(len(sig) - 1):
            t, t_next = tf(sig[i]), tf(sig[i + 1])
            h = t_next - t
            s = t + h * self.r
            fac = 1 / (2 * self.r)
            st = prog.step()
            st.denoise(DN, X, _f(sig[i]))
            # step 1: to sigma(s)
            sd, su = _ancestral_step(sg(t), sg(s), self.eta)
            s_ = tf(torch.as_tensor(sd))
            terms = [(X, _f(sg(s_) / sg(t))), (DN, -_f((t - s_).expm1()))]
            if _f(su) != 0.0:
                st.randn(NZ)  # A = (W(sigma_i) - W(sigma_s)) / sqrt(d1); the reference's draw is (W(sigma_s) - W(sigma_i)) / sqrt(d1) = -A
                terms.append((NZ, -_f(self.s_noise * su)))
            st.lin(X2, terms)
            st.denoise(DN2, X2, _f(sg(s)))
            # step 2: to sigma(t_next), on the mixed estimate
            sd, su2 = _ancestral_step(sg(t), sg(t_next), self.eta)
            t_next_ = tf(torch.as_tensor(sd))
            e = _f((t - t_next_).expm1())
            terms = [(X, _f(sg(t_next_) / sg(t))), (DN, -e * (1 - fac)), (DN2, -e * fac)]
            if _f(su2) != 0.0:
                d1, d2 = _f(sg(t) - sg(s)), _f(sg(s) - sg(t_next))
                amp = _f(self.s_noise * su2)
                if _f(su) == 0.0:
                    st.randn(NZ)
                st.randn(XH)  # B = (W(sigma_s) - W(sigma_next)) / sqrt(d2)
                terms += [(NZ, -amp * math.sqrt(d1 / (d1 + d2))), (XH, -amp * math.sqrt(d2 / (d1 + d2)))]
            st.lin(X, terms)
        return prog


class DPMPP2MSDE(_BrownianSDE):
    """DPM-Solver++(2M) SDE, models/sample.py:451-518.  SAMPLER_OPTIONS: SOLVER ('heun' | 'midpoint'), ETA (0), S_NOISE (1)."""

    def __init__(self, config):
        super().__init__(config)
        self.solver_type = self.sample_config.get("SOLVER", "heun")
        if self.solver_type not in {"heun", "midpoint"}:
            raise ValueError("'SOLVER' must be 'heun' or 'midpoint'")

    def build(self, model, num_steps, sample_offset) -> Program:
        sig = self.setup_sigmas(model, num_steps)
        prog = Program(6, _f(sig[0]))
        OLD = X2
        h_last = None
        for i in range(len(sig) - 1):
            st = prog.step()
            st.denoise(DN, X, _f(sig[i]))
            if sig[i + 1] == 0:
                st.lin(X, [(DN, 1.0)])
            else:
                t, s = -sig[i].log(), -sig[i + 1].log()
                h = s - t
                eta_h = self.eta * h
                a, b = _f(sig[i + 1] / sig[i] * (-eta_h).exp()), _f((-h - eta_h).expm1().neg())
                c = 0.0
                if h_last is not None:
                    r = h_last / h
                    if self.solver_type == "heun":
                        c = _f(((-h - eta_h).expm1().neg() / (-h - eta_h) + 1) * (1 / r))
                    else:
                        c = _f(0.5 * (-h - eta_h).expm1().neg() * (1 / r))
                terms = [(X, a), (DN, b + c), (OLD, -c)]
                if self.eta:
                    st.randn(NZ)  # A over [sigma_next, sigma_i]; the reference's draw is -A
                    terms.append((NZ, -_f(sig[i + 1] * (-2 * eta_h).expm1().neg().sqrt() * self.s_noise)))
                st.lin(X, terms)
                h_last = h
            st.lin(OLD, [(DN, 1.0)])
        return prog


class DPMPP3MSDE(_BrownianSDE):
    """DPM-Solver++(3M) SDE, models/sample.py:521-574.  SAMPLER_OPTIONS: ETA (0), S_NOISE (1)."""

    def build(self, model, num_steps, sample_offset) -> Program:
        sig = self.setup_sigmas(model, num_steps)
        prog = Program(8, _f(sig[0]))
        D1, D2 = X2, DN2  # denoised_1, denoised_2
        h_1 = h_2 = None
        for i in range(len(sig) - 1):
            st = prog.step()
            st.denoise(DN, X, _f(sig[i]))
            if sig[i + 1] == 0:
                st.lin(X, [(DN, 1.0)])
                h = None
            else:
                t, s = -sig[i].log(), -sig[i + 1].log()
                h = s - t
                h_eta = h * (self.eta + 1)
                cx, cd, c1, c2 = _f(torch.exp(-h_eta)), _f((-h_eta).expm1().neg()), 0.0, 0.0
                if h_2 is not None:
                    r0, r1 = h_1 / h, h_2 / h
                    phi_2 = h_eta.neg().expm1() / h_eta + 1
                    phi_3 = phi_2 / h_eta - 0.5
                    k = r0 / (r0 + r1)
                    # x += phi_2 d1 - phi_3 d2 with d1_0 = (D - D1) / r0, d1_1 = (D1 - D2) / r1,
                    # d1 = d1_0 + (d1_0 - d1_1) k, d2 = (d1_0 - d1_1) / (r0 + r1)
                    P = phi_2 * (1 + k) - phi_3 / (r0 + r1)   # coefficient of d1_0
                    Q = -phi_2 * k + phi_3 / (r0 + r1)        # coefficient of d1_1
                    cd += _f(P / r0)
                    c1 = _f(-P / r0 + Q / r1)
                    c2 = _f(-Q / r1)
                elif h_1 is not None:
                    r = h_1 / h
                    phi_2 = h_eta.neg().expm1() / h_eta + 1
                    cd += _f(phi_2 / r)
                    c1 = -_f(phi_2 / r)
                terms = [(X, cx), (DN, cd), (D1, c1), (D2, c2)]
                nz = _f(sig[i + 1] * (-2 * h * self.eta).expm1().neg().sqrt() * self.s_noise)
                if nz != 0.0:
                    st.randn(NZ)  # A over [sigma_next, sigma_i]; the reference's draw is -A
                    terms.append((NZ, -nz))
                st.lin(X, terms)
            st.lin(D2, [(D1, 1.0)])
            st.lin(D1, [(DN, 1.0)])
            h_1, h_2 = h, h_1
        return prog


BespokeNonStationary = _unavailable("BespokeNonStationary", "needs a trained theta file (SAMPLER_PATH)")
