"""CPU oracle for the noise-schedule sampler loop.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Restates, for the branches reachable from ``GaussianDiffusion.forward``:

* beta schedule / cumulative products   -- fish_diffusion/archs/diffsinger/diffusions/diffusion.py:18-31,70-88
* sampler driver                         -- diffusion.py:196-313, norm/denorm :315-319, q_sample :120-127
* DDPM ancestral ("naive") step          -- noise_predictor.py:19-104
* PLMS step + Adams-Bashforth blends     -- noise_predictor.py:107-148, driver diffusion.py:269-311
* discrete VP schedule + interpolation   -- uni_pc.py:81-99,124-161,826-875
* UniPC bh2 / order 2 / multistep / time_uniform / data-prediction
                                         -- uni_pc.py:583-701 (update), :703-818 (sample), :340-351, :200-283

The denoiser is passed in as a callable ``eps = denoise(x[B,M,T], t[B], cond[B,E,T], x_masks, cond_masks)``
so the same loop can drive the CPU oracle WaveNet or (in tests) be compared with the device path.
All random draws are explicit inputs (the reference draws them from torch's global RNG at
diffusion.py:222,232 and noise_predictor.py:101); make_golden.py pins the draw order.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch


# ------------------------------------------------------------------ schedule (float64 numpy, as the reference)
def beta_schedule(mode="linear", timesteps=1000, max_beta=0.01, s=0.008) -> np.ndarray:
    """diffusion.py:18-31."""
    if mode == "linear":
        return np.linspace(1e-4, max_beta, timesteps)
    if mode == "cosine":
        n = timesteps + 1
        x = np.linspace(0, n, n)
        ac = np.cos(((x / n) + s) / (1 + s) * np.pi * 0.5) ** 2
        ac = ac / ac[0]
        return np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    raise NotImplementedError(mode)


def f32(a) -> torch.Tensor:
    return torch.tensor(a, dtype=torch.float32)


# ------------------------------------------------------------------ naive (DDPM ancestral)
class NaiveTables:
    """noise_predictor.py:29-71 -- float64 numpy, cast to fp32 tensors.  In the reference these are `register_buffer`s: `from_buffers`
    takes them as a (loaded) state dict of NaiveNoisePredictor instead of recomputing them from the betas."""

    def __init__(self, betas: np.ndarray):
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.sqrt_recip = f32(np.sqrt(1.0 / ac))
        self.sqrt_recipm1 = f32(np.sqrt(1.0 / ac - 1))
        var = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.logvar = f32(np.log(np.maximum(var, 1e-20)))
        self.coef1 = f32(betas * np.sqrt(ac_prev) / (1.0 - ac))
        self.coef2 = f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))
        self.clip_min, self.clip_max = -1.0, 1.0

    @classmethod
    def from_buffers(cls, sd: dict) -> "NaiveTables":
        self = cls.__new__(cls)
        self.sqrt_recip, self.sqrt_recipm1 = sd["sqrt_recip_alphas_cumprod"].float(), sd["sqrt_recipm1_alphas_cumprod"].float()
        self.logvar = sd["posterior_log_variance_clipped"].float()
        self.coef1, self.coef2 = sd["posterior_mean_coef1"].float(), sd["posterior_mean_coef2"].float()
        self.clip_min, self.clip_max = float(sd["clip_min"]), float(sd["clip_max"])
        return self


def naive_step(tb: NaiveTables, x, t: int, eps, noise):
    """noise_predictor.py:73-104."""
    x0 = tb.sqrt_recip[t] * x - tb.sqrt_recipm1[t] * eps
    x0 = torch.clamp(x0, min=tb.clip_min, max=tb.clip_max)
    mean = tb.coef1[t] * x0 + tb.coef2[t] * x
    nonzero = 1.0 if t > 0 else 0.0
    return mean + nonzero * (0.5 * tb.logvar[t]).exp() * noise


# ------------------------------------------------------------------ PLMS
def plms_x_pred(ac: torch.Tensor, x, eps, t: int, t_prev: int):
    """noise_predictor.py:118-131 (ac = fp32 alphas_cumprod)."""
    a_t, a_prev = ac[t], ac[t_prev]
    a_t_sq, a_prev_sq = a_t.sqrt(), a_prev.sqrt()
    delta = (a_prev - a_t) * (
        (1 / (a_t_sq * (a_t_sq + a_prev_sq))) * x
        - 1 / (a_t_sq * (((1 - a_prev) * a_t).sqrt() + ((1 - a_t) * a_prev).sqrt())) * eps
    )
    return x + delta


def plms_blend(eps, hist: List[torch.Tensor], eps_prev_first=None):
    """noise_predictor.py:133-148; stage = len(hist)."""
    n = len(hist)
    if n == 0:
        return (eps + eps_prev_first) / 2
    if n == 1:
        return (eps * 3 - hist[-1]) / 2
    if n == 2:
        return (eps * 23 - hist[-1] * 16 + hist[-2] * 5) / 12
    return (eps * 55 - hist[-1] * 59 + hist[-2] * 37 - hist[-3] * 9) / 24


# ------------------------------------------------------------------ discrete VP schedule for UniPC
class DiscreteVP:
    """uni_pc.py:81-99 (schedule='discrete', betas as float64 tensor -> fp32 arrays), :124-161."""

    def __init__(self, betas: np.ndarray):
        b = torch.from_numpy(np.asarray(betas, dtype=np.float64))
        self.log_alpha = (0.5 * torch.log(1 - b).cumsum(dim=0)).to(torch.float32)
        self.N = len(self.log_alpha)
        self.t_knots = torch.linspace(0.0, 1.0, self.N + 1)[1:].to(torch.float32)

    def log_mean_coeff(self, t: torch.Tensor) -> torch.Tensor:
        """Piecewise-linear interpolation with linear extrapolation outside the knots
        (uni_pc.py:826-875).  t: [n] fp32 -> [n].  At a knot the reference's sort-based
        lookup lands on the segment whose *upper* end is that knot (pinned in make_golden.py)."""
        xp, yp, K = self.t_knots, self.log_alpha, self.N
        # segment index j: interpolate between knots j and j+1
        idx = torch.searchsorted(xp, t, right=False)  # first knot >= t
        j = torch.clamp(idx - 1, 0, K - 2)
        x0, x1, y0, y1 = xp[j], xp[j + 1], yp[j], yp[j + 1]
        return y0 + (t - x0) * (y1 - y0) / (x1 - x0)

    def alpha(self, t):
        return torch.exp(self.log_mean_coeff(t))

    def sigma(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.log_mean_coeff(t)))

    def lam(self, t):
        la = self.log_mean_coeff(t)
        return la - 0.5 * torch.log(1.0 - torch.exp(2.0 * la))


def unipc_sample(eps_model: Callable, x: torch.Tensor, betas: np.ndarray, steps: int, trace: Optional[list] = None):
    """UniPC(variant='bh2').sample(order=2, skip_type='time_uniform', method='multistep').
    ``eps_model(x, t_input[B])`` returns the noise prediction; t_input = (t - 1/N) * N (uni_pc.py:220-223)."""
    ns = DiscreteVP(betas)
    B = x.shape[0]
    ts = torch.linspace(1.0, 1.0 / ns.N, steps + 1)  # uni_pc.py:719-722, 371-372
    assert steps >= 2

    def x0_pred(xx, t):  # uni_pc.py:340-351 (no thresholding)
        t1 = t.view(-1)
        eps = eps_model(xx, ((t1 - 1.0 / ns.N) * ns.N).expand(B))
        if trace is not None:
            trace.append(eps)
        return (xx - ns.sigma(t1) * eps) / ns.alpha(t1)

    t_hist = [ts[0].view(-1)]
    m_hist = [x0_pred(x, ts[0])]
    for step in range(1, steps + 1):
        t = ts[step].view(-1)
        order = step if step < 2 else min(2, steps + 1 - step)
        use_corrector = step < steps
        # ---- uni_pc.py:583-701, predict_x0 branch
        t0, m0 = t_hist[-1], m_hist[-1]
        lam0, lam_t = ns.lam(t0), ns.lam(t)
        sig0, sig_t = ns.sigma(t0), ns.sigma(t)
        alpha_t = torch.exp(ns.log_mean_coeff(t))
        h = lam_t - lam0
        rks, D1 = [], None
        if order == 2:
            rk = (ns.lam(t_hist[-2]) - lam0) / h
            rks.append(rk)
            D1 = (m_hist[-2] - m0) / rk
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        R, b = torch.stack(R), torch.cat(b)
        x_base = sig_t / sig0 * x - alpha_t * h_phi_1 * m0
        pred_res = 0.5 * D1 if D1 is not None else 0
        x_t = x_base - alpha_t * B_h * pred_res
        m_t = None
        if use_corrector:
            rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
            m_t = x0_pred(x_t, t)
            corr_res = rhos_c[0] * D1 if D1 is not None else 0
            x_t = x_base - alpha_t * B_h * (corr_res + rhos_c[-1] * (m_t - m0))
        x = x_t
        # ---- history (uni_pc.py:758-804)
        if step < 2:
            t_hist.append(t)
            m_hist.append(m_t)
        else:
            t_hist = [t_hist[1], t]
            if step < steps:
                m_hist = [m_hist[1], m_t]
    return x


# ------------------------------------------------------------------ driver
def norm_spec(x, spec_min, spec_max):
    return (x - spec_min) / (spec_max - spec_min) * 2 - 1


def denorm_spec(x, spec_min, spec_max):
    return (x + 1) / 2 * (spec_max - spec_min) + spec_min


def diffusion_sample(denoise: Callable, features: torch.Tensor, *, x_init: torch.Tensor,
                     sampler_interval=10, predictor: Optional[str] = None, step_noise: Optional[torch.Tensor] = None,
                     noise_schedule="linear", timesteps=1000, max_beta=0.01, s=0.008,
                     spec_min=(-5.0,), spec_max=(0.0,), skip_steps=0,
                     x_masks=None, cond_masks=None, trace: Optional[list] = None,
                     naive_buffers: Optional[dict] = None, plms_alphas_cumprod: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GaussianDiffusion.forward (diffusion.py:196-313).

    features [B,T,E]; x_init [B,M,T] is the tensor the reference would hold right before the loop
    (randn at :222, or q_sample(norm_spec(original_mel)) at :223-232 -- build it with ``q_sample``);
    step_noise [n_steps,B,M,T] for the naive predictor.  Returns mel [B,T,M].
    `naive_buffers` / `plms_alphas_cumprod`: the predictor modules' buffers as a checkpoint holds them (noise_predictor.py:29-71,115);
    default: computed from the schedule, as the constructors do.
    """
    betas = beta_schedule(noise_schedule, timesteps, max_beta, s)
    if predictor is None:
        predictor = "naive" if sampler_interval == 1 else "unipc"
    predictor = predictor.lower()
    cond = features.transpose(1, 2)
    x = x_init
    chunks = list(range(0, timesteps - skip_steps, sampler_interval))[::-1]
    smin = torch.tensor(spec_min, dtype=torch.float32).view(1, 1, -1)
    smax = torch.tensor(spec_max, dtype=torch.float32).view(1, 1, -1)
    B = x.shape[0]

    def call(xx, t, masked=True):
        tt = t if torch.is_tensor(t) else torch.full((1,), t, dtype=torch.long)
        eps = denoise(xx, tt, cond, x_masks if masked else None, cond_masks if masked else None)
        if trace is not None:
            trace.append(eps)
        return eps

    if predictor == "naive":
        tb = NaiveTables(betas) if naive_buffers is None else NaiveTables.from_buffers(naive_buffers)
        for i, t in enumerate(chunks):
            eps = call(x, t)
            x = naive_step(tb, x, t, eps, step_noise[i])
    elif predictor == "unipc":
        steps = timesteps // sampler_interval  # noise_predictor.py:187 (total_N, NOT timesteps-skip_steps)
        x = unipc_sample(lambda xx, t: denoise(xx, t, cond, x_masks, cond_masks), x, betas, steps, trace)
    elif predictor == "plms":
        ac = f32(np.cumprod(1.0 - betas, axis=0)) if plms_alphas_cumprod is None else plms_alphas_cumprod.float()
        hist: List[torch.Tensor] = []
        for t in chunks:
            eps = call(x, t)
            t_prev = max(t - sampler_interval, 0)
            if len(hist) == 0:
                x_pred = plms_x_pred(ac, x, eps, t, t_prev)
                eps_prev = call(x_pred, t_prev, masked=False)  # diffusion.py:285 -- no masks
                eps_prime = plms_blend(eps, hist, eps_prev)
            else:
                eps_prime = plms_blend(eps, hist)
            hist = (hist + [eps])[-3:]
            x = plms_x_pred(ac, x, eps_prime, t, t_prev)
    else:
        raise NotImplementedError(f"Unknown noise predictor: {predictor}")
    return denorm_spec(x.transpose(1, 2), smin, smax)


def q_sample(x_start, t: int, noise, betas: np.ndarray):
    """diffusion.py:120-127."""
    ac = np.cumprod(1.0 - betas, axis=0)
    return f32(np.sqrt(ac))[t] * x_start + f32(np.sqrt(1.0 - ac))[t] * noise


def ddpm_noise_stream(seed, B, M, T, n_steps):
    """The draws GaussianDiffusion.forward makes from torch's global RNG in a `noise_predictor="naive"` run, in order
    (diffusion.py:222 `torch.randn(shape)`, then one `torch.randn_like(x)` per step, noise_predictor.py:101): yields x_T, then the
    n_steps step noises.  Restores nothing: callers own the global RNG while iterating."""
    torch.manual_seed(seed)
    yield torch.randn(B, M, T)
    for _ in range(n_steps):
        yield torch.randn(B, M, T)


def ddpm_noise(seed, B, M, T, n_steps):
    it = ddpm_noise_stream(seed, B, M, T, n_steps)
    x_init = next(it)
    return x_init, torch.stack(list(it))
