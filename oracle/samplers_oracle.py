"""CPU restatement of the reference's other samplers (calodiffusion/models/sample.py), written as plain update loops over a
``denoise(x, sigma)`` callable.  TEST INFRASTRUCTURE ONLY (same rules as torch_oracle.py): the product builds these samplers as
step programs for cd_sampler_run (calodiffusion_amd/sample.py); this file states the same mathematics the straightforward way,
pinned against trajectories of the reference's own classes (tests/golden/samplers_*.npz, tests/test_oracle_golden.py).

``noise`` is an iterator of unit-normal tensors, consumed in the reference's draw order.  Schedules are fp32 like the
reference's 0-dim tensor arithmetic.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Iterator, List, Optional

import numpy as np
import torch

Tensor = torch.Tensor
Denoise = Callable[[Tensor, Tensor], Tensor]


def karras(num_steps: int, sample_offset: int = 0, sigma_min=0.002, sigma_max=80.0, rho=7) -> Tensor:
    """EDMAbstract.setup (sample.py:667-685)."""
    i = torch.arange(num_steps, dtype=torch.float32)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros_like(t[:1])])[sample_offset:]


def _churn(t_cur, num_steps, s_churn, s_min, s_max, gamma_on=None):
    g = min(s_churn / num_steps, np.sqrt(2) - 1) if gamma_on is None else gamma_on
    gamma = g if s_min <= t_cur <= s_max else 0
    return torch.as_tensor(t_cur + gamma * t_cur)


def edm_loop(kind: str, denoise: Denoise, start: Tensor, num_steps: int, noise: Iterator[Tensor], noisy=False, sample_offset=0,
             s_noise=1.003, s_min=0.01):
    """EDMAbstract.for_loop with Euler / Heun / DPM2.in_loop_sampler (sample.py:631-662, 771-851).  Heun's second slope uses
    x_cur -- `self.x_next` still holds it when in_loop_sampler runs (sample.py:820)."""
    s_churn, s_max = (40, 50) if noisy else (0, 1)
    t = karras(num_steps, sample_offset)
    x_next = start.float() * t[0]
    xs, x0s = [], []
    for t_cur, t_next in zip(t[:-1], t[1:]):
        x_cur = x_next
        t_hat = _churn(t_cur, num_steps, s_churn, s_min, s_max)
        x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * s_noise * next(noise)
        den = denoise(x_hat, t_hat)
        d_cur = (x_hat - den) / t_hat
        h = t_next - t_hat
        if kind == "euler":
            x_next = x_hat + h * d_cur
        elif kind == "heun":
            x_prime = x_hat + h * d_cur
            d_prime = (x_cur - denoise(x_prime, t_hat + h)) / t_next
            x_next = x_hat + h * (0.5 * d_cur + 0.5 * d_prime)
        elif kind == "dpm2":
            t_mid = t_hat.log().lerp(t_next.log(), 0.5).exp()
            x_2 = x_hat + d_cur * (t_mid - t_hat)
            d_2 = (x_2 - denoise(x_2, t_mid)) / t_mid
            x_next = x_hat + h * d_2
        else:
            raise KeyError(kind)
        xs.append(x_cur)
        x0s.append(den)
    return x_next, xs, x0s


def lms(denoise: Denoise, start: Tensor, num_steps: int, order=4, sample_offset=0):
    """LMS.sampler (sample.py:741-768); coefficients by scipy quadrature like utils/sampling.py:77-92."""
    from scipy import integrate
    t = karras(num_steps, sample_offset)
    tn = t.double().numpy()

    def coeff(cur, i, j):
        def fn(tau):
            prod = 1.0
            for k in range(cur):
                if j != k:
                    prod *= (tau - tn[i - k]) / (tn[i - j] - tn[i - k])
            return prod
        return integrate.quad(fn, tn[i], tn[i + 1], epsrel=1e-4)[0]

    x = start.float() * t[0]
    ds: List[Tensor] = []
    for i, t_cur in enumerate(t[:-1]):
        d = (x - denoise(x, t_cur)) / t_cur
        ds.append(d)
        if len(ds) > order:
            ds.pop(0)
        cur = min(i + 1, order)
        x = x + sum(coeff(cur, i, j) * dj for j, dj in zip(range(cur), reversed(ds)))
    return x


def restart(denoise: Denoise, start: Tensor, num_steps: int, noise: Iterator[Tensor], restart_list: Dict, noisy=False,
            s_noise=1.003, s_min=0.01, restart_gamma=0.05, rho=7):
    """Restart.sampler / restart_loop (sample.py:871-954).  `index + 1 in restart_list` is tested with an int key."""
    s_churn, s_max = (40, 50) if noisy else (0, 1)
    t = karras(num_steps)
    x_next = start.float() * t[0]
    x0s = []
    for index, (t_cur, t_next) in enumerate(zip(t[:-1], t[1:])):
        t_hat = _churn(t_cur, num_steps, s_churn, s_min, s_max)
        x_hat = x_next + (t_hat ** 2 - t_cur ** 2).sqrt() * s_noise * next(noise)
        den = denoise(x_hat, t_hat)
        x0s.append(den)
        x_next = x_hat + (t_next - t_hat) * (x_hat - den) / t_hat
        if index + 1 not in restart_list:
            continue
        n_restart, k_rep, _, t_max = restart_list[index + 1]
        for _ in range(int(k_rep)):
            i = torch.arange(int(n_restart), dtype=torch.float32)
            nt = (t_max ** (1 / rho) + i / (int(n_restart) - 1) * (t[index + 1] ** (1 / rho) - t_max ** (1 / rho))) ** rho
            x_next = x_next + next(noise) * (nt[0] ** 2 - nt[-1] ** 2).sqrt() * s_noise
            for j, (tc, tn) in enumerate(zip(nt[:-1], nt[1:])):
                th = _churn(tc, num_steps, s_churn, s_min, s_max, gamma_on=restart_gamma)
                xh = x_next + (th ** 2 - tc ** 2).sqrt() * s_noise * next(noise)
                d_cur = (xh - denoise(xh, th)) / th
                x_next = xh + (tn - th) * d_cur
                if j < len(nt) - 2 or nt[-1] != 0:
                    d_prime = (x_next - denoise(x_next, tn)) / tn
                    x_next = xh + (tn - th) * (0.5 * d_cur + 0.5 * d_prime)
    return x_next, x0s


def model_sigmas(tables, num_steps: int) -> Tensor:
    """DPM.create_sigmas (sample.py:146-153): the cosine-schedule noise levels, largest first."""
    return torch.tensor([tables.sqrt_one_minus_alphas_cumprod[num_steps - k - 1] / tables.sqrt_alphas_cumprod[num_steps - k - 1]
                         for k in range(num_steps)])


def dpmpp2m(denoise: Denoise, start: Tensor, sig: Tensor):
    """DPMPP2M.sample (sample.py:424-449)."""
    x = start * sig[0]
    old = None
    for i in range(len(sig) - 1):
        den = denoise(x, sig[i])
        t, t_next = -sig[i].log(), -sig[i + 1].log()
        h = t_next - t
        if old is None or sig[i + 1] == 0:
            d = den
        else:
            r = (t - (-sig[i - 1].log())) / h
            d = (1 + 1 / (2 * r)) * den - (1 / (2 * r)) * old
        x = (sig[i + 1] / sig[i]) * x - (-h).expm1() * d
        old = den
    return x


def dpmpp2s(denoise: Denoise, start: Tensor, sig: Tensor, noise: Iterator[Tensor], eta=0.0, s_noise=1.0):
    """DPMPP2S.sample (sample.py:312-344), the ancestral noise added once after the loop as there."""
    x = start * sig[0]
    up = 0.0
    for i in range(len(sig) - 1):
        den = denoise(x, sig[i])
        if not eta:
            down, up = sig[i + 1], 0.0
        else:
            up = min(sig[i + 1], eta * (sig[i + 1] ** 2 * (sig[i] ** 2 - sig[i + 1] ** 2) / sig[i] ** 2) ** 0.5)
            down = (sig[i + 1] ** 2 - up ** 2) ** 0.5
        t, t_next = -sig[i].log(), -torch.as_tensor(down).log()
        h = t_next - t
        s = t + 0.5 * h
        x_2 = ((-s).exp() / (-t).exp()) * x - (-h * 0.5).expm1() * den
        den_2 = denoise(x_2, (-s).exp())
        x = ((-t_next).exp() / (-t).exp()) * x - (-h).expm1() * den_2
    if sig[-1] > 0:
        x = x + next(noise) * s_noise * up
    return x


class BrownianPath:
    """Stand-in for sampling.BrownianTreeNoiseSampler (utils/sampling.py:327-382; torchsde is not installed here, so the reference's
    SDE samplers cannot run and these three restatements are PARITY UNPINNED: no reference trajectory exists for them in this
    tree).  The reference's sampler returns noise(sigma, sigma') = (W(sigma') - W(sigma)) / sqrt|sigma' - sigma| for a Brownian path
    W over sigma (BatchedBrownianTree.__call__ sorts the pair and multiplies by the sort sign, which gives exactly this difference).
    Here W is built on the points the sampler asks for, walking DOWN the schedule: a new point p below the lowest known point L
    gets W(p) = W(L) - sqrt(L - p) xi with xi the next unit normal of `noise` -- xi is the normalised increment over [p, L]."""

    def __init__(self, noise: Iterator[Tensor], sigma_top):
        self.noise = noise
        self.pts = [(float(sigma_top), 0.0)]  # (sigma, W), sigma descending

    def w(self, sigma):
        sigma = float(sigma)
        for p, wv in self.pts:
            if p == sigma:
                return wv
        low, wl = self.pts[-1]
        assert sigma < low, "BrownianPath: points must be requested walking down the schedule"
        wv = wl - math.sqrt(low - sigma) * next(self.noise)
        self.pts.append((sigma, wv))
        return wv

    def __call__(self, sigma, sigma_next):
        w0 = self.w(sigma)
        return (self.w(sigma_next) - w0) / math.sqrt(abs(float(sigma_next) - float(sigma)))


def _ancestral_step(sigma_from, sigma_to, eta):
    """get_ancestral_step (utils/sampling.py:31-41)."""
    if not eta:
        return sigma_to, 0.0
    up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    return (sigma_to ** 2 - up ** 2) ** 0.5, up


def dpmpp_sde(denoise: Denoise, start: Tensor, sig: Tensor, noise: Iterator[Tensor], eta=0.0, s_noise=1.0, r=0.5):
    """DPMPPSDE.sample (sample.py:363-416).  A noise term whose amplitude is exactly zero is skipped (the reference multiplies its
    draw by 0.0): with eta = 0 nothing is drawn."""
    ns = BrownianPath(noise, sig[0])
    x = start * sig[0]
    for i in range(len(sig) - 1):
        den = denoise(x, sig[i])
        t, t_next = -sig[i].log(), -sig[i + 1].log()
        h = t_next - t
        s = t + h * r
        fac = 1 / (2 * r)
        sd, su = _ancestral_step((-t).exp(), (-s).exp(), eta)
        s_ = -torch.as_tensor(sd).log()
        x_2 = ((-s_).exp() / (-t).exp()) * x - (t - s_).expm1() * den
        # (path points: the schedule's own sigma_i / sigma_next and sigma(s); the reference passes sigma_fn(time_fn(sigma_i)), the same
        # number to rounding -- a point one ulp off would be a new, spurious increment of the path here)
        sig_s = (-s).exp()
        if float(su) != 0.0:
            x_2 = x_2 + ns(sig[i], sig_s) * s_noise * su
        den_2 = denoise(x_2, sig_s)
        sd, su = _ancestral_step((-t).exp(), (-t_next).exp(), eta)
        t_next_ = -torch.as_tensor(sd).log()
        den_d = (1 - fac) * den + fac * den_2
        x = ((-t_next_).exp() / (-t).exp()) * x - (t - t_next_).expm1() * den_d
        if float(su) != 0.0:
            ns.w(sig[i]), ns.w(sig_s)  # (the path passes through sigma(s) whether or not step 1 drew there)
            x = x + ns(sig[i], sig[i + 1]) * s_noise * su
    return x


def dpmpp_2m_sde(denoise: Denoise, start: Tensor, sig: Tensor, noise: Iterator[Tensor], eta=0.0, s_noise=1.0, solver="heun"):
    """DPMPP2MSDE.sample (sample.py:468-518)."""
    ns = BrownianPath(noise, sig[0])
    x = start * sig[0]
    old, h_last = None, None
    for i in range(len(sig) - 1):
        den = denoise(x, sig[i])
        if sig[i + 1] == 0:
            x = den
        else:
            t, s = -sig[i].log(), -sig[i + 1].log()
            h = s - t
            eta_h = eta * h
            x = sig[i + 1] / sig[i] * (-eta_h).exp() * x + (-h - eta_h).expm1().neg() * den
            if old is not None:
                rr = h_last / h
                if solver == "heun":
                    x = x + ((-h - eta_h).expm1().neg() / (-h - eta_h) + 1) * (1 / rr) * (den - old)
                else:
                    x = x + 0.5 * (-h - eta_h).expm1().neg() * (1 / rr) * (den - old)
            if eta:
                x = x + ns(sig[i], sig[i + 1]) * sig[i + 1] * (-2 * eta_h).expm1().neg().sqrt() * s_noise
        old = den
        h_last = h
    return x


def dpmpp_3m_sde(denoise: Denoise, start: Tensor, sig: Tensor, noise: Iterator[Tensor], eta=0.0, s_noise=1.0):
    """DPMPP3MSDE.sample (sample.py:530-574); the noise term is skipped where its amplitude is exactly zero (eta = 0)."""
    ns = BrownianPath(noise, sig[0])
    x = start * sig[0]
    den_1 = den_2 = None
    h_1 = h_2 = None
    for i in range(len(sig) - 1):
        den = denoise(x, sig[i])
        if sig[i + 1] == 0:
            x = den
            h = None
        else:
            t, s = -sig[i].log(), -sig[i + 1].log()
            h = s - t
            h_eta = h * (eta + 1)
            x = torch.exp(-h_eta) * x + (-h_eta).expm1().neg() * den
            if h_2 is not None:
                r0, r1 = h_1 / h, h_2 / h
                d1_0, d1_1 = (den - den_1) / r0, (den_1 - den_2) / r1
                d1 = d1_0 + (d1_0 - d1_1) * r0 / (r0 + r1)
                d2 = (d1_0 - d1_1) / (r0 + r1)
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                phi_3 = phi_2 / h_eta - 0.5
                x = x + phi_2 * d1 - phi_3 * d2
            elif h_1 is not None:
                rr = h_1 / h
                d = (den - den_1) / rr
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                x = x + phi_2 * d
            amp = sig[i + 1] * (-2 * h * eta).expm1().neg().sqrt() * s_noise
            if float(amp) != 0.0:
                x = x + ns(sig[i], sig[i + 1]) * amp
        den_1, den_2 = den, den_1
        h_1, h_2 = h, h_1
    return x


def dpm_fast(denoise: Denoise, start: Tensor, sig: Tensor, nfe: int):
    """DPM.sample -> DPMSolver.dpm_solver_fast with eta = 0 (sample.py:164-177, utils/sampling.py:412-506)."""
    x = start * sig[0]
    t_start, t_end = -sig[0].log(), -sig[-1].log()
    m = math.floor(nfe / 3) + 1
    ts = torch.linspace(t_start, t_end, m + 1)
    orders = [3] * (m - 2) + [2, 1] if nfe % 3 == 0 else [3] * (m - 1) + [nfe % 3]
    sg = lambda tt: tt.neg().exp()  # noqa: E731
    eps_of = lambda xx, tt: (xx - denoise(xx, sg(tt))) / sg(tt)  # noqa: E731
    for i, order in enumerate(orders):
        t, tn = ts[i], ts[i + 1]
        h = tn - t
        eps = eps_of(x, t)
        if order == 1:
            x = x - sg(tn) * h.expm1() * eps
            continue
        r1 = 1 / 2 if order == 2 else 1 / 3
        s1 = t + r1 * h
        u1 = x - sg(s1) * (r1 * h).expm1() * eps
        eps_r1 = eps_of(u1, s1)
        if order == 2:
            x = x - sg(tn) * h.expm1() * eps - sg(tn) / (2 * r1) * h.expm1() * (eps_r1 - eps)
            continue
        r2 = 2 / 3
        s2 = t + r2 * h
        u2 = x - sg(s2) * (r2 * h).expm1() * eps - sg(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1) * (eps_r1 - eps)
        eps_r2 = eps_of(u2, s2)
        x = x - sg(tn) * h.expm1() * eps - sg(tn) / r2 * (h.expm1() / h - 1) * (eps_r2 - eps)
    return x


def consistency(denoise: Denoise, start: Tensor, tables_n, consis_nsteps: int, num_steps: int, noise: Iterator[Tensor],
                sigma_min=0.002):
    """Consistency.__call__ + sampling.sample_consis (sample.py:968-1011, utils/sampling.py:1143-1173);
    `tables_n` = the schedule tables for consis_nsteps steps."""
    n = consis_nsteps
    idx = [0, int(round(n * 0.5)), int(round(n * 0.7)), int(round(n * 0.9)), int(round(n * 0.95))]
    t_all = model_sigmas(tables_n, n)
    t_steps = torch.tensor([t_all[i] for i in idx[:num_steps]]) if num_steps > 1 else torch.tensor([t_all[0]])
    sig = torch.cat([t_steps, torch.zeros_like(t_steps[:1])])
    x = start * sig[0]
    xs, x0 = [], None
    for s_cur, s_next in zip(sig[:-1], sig[1:]):
        x0 = denoise(x, s_cur)
        s_next = torch.clip(s_next, sigma_min, None)
        x = x0 + next(noise) * torch.sqrt(s_next ** 2 - sigma_min ** 2) if s_next > sigma_min else x0
        xs.append(x)
    return x, xs, x0
