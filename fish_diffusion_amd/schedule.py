"""Host-side scalar tables for the sampler loop (the rows of `fdx_sampler_run`, include/fishdx.h).

Everything here is O(n_steps) scalar arithmetic; the tensors never leave the device.  The scalars are
computed with torch fp32 ops on one-element CPU tensors in the same order as the reference evaluates
them (diffusion.py:18-31,70-88; noise_predictor.py:29-71,118-131; uni_pc.py:81-99,124-161,583-663), so
that the device update rules see bit-identical coefficients.  Tables are cached per configuration.
"""
from __future__ import annotations

import functools
from typing import Tuple

import numpy as np
import torch

from ._lib import FDX_ROW, SAMPLER_NAIVE, SAMPLER_PLMS, SAMPLER_UNIPC

KINDS = {"naive": SAMPLER_NAIVE, "unipc": SAMPLER_UNIPC, "plms": SAMPLER_PLMS}


def make_betas(noise_schedule="linear", timesteps=1000, max_beta=0.01, s=0.008) -> np.ndarray:
    """float64 numpy, diffusion.py:18-31."""
    if noise_schedule == "linear":
        return np.linspace(1e-4, max_beta, timesteps)
    if noise_schedule == "cosine":
        steps = timesteps + 1
        grid = np.linspace(0, steps, steps)
        acp = np.cos(((grid / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
        acp = acp / acp[0]
        return np.clip(1 - (acp[1:] / acp[:-1]), a_min=0, a_max=0.999)
    raise NotImplementedError


def _f32(a):
    return torch.tensor(a, dtype=torch.float32)


def timestep_chunks(timesteps: int, skip_steps: int, interval: int):
    """diffusion.py:234-240: arange(0, T - skip, interval) flipped."""
    return list(range(0, timesteps - skip_steps, interval))[::-1]


# ------------------------------------------------------------------ naive (DDPM ancestral)
def naive_buffers(betas: np.ndarray) -> dict:
    """The fp32 buffers NaiveNoisePredictor registers (noise_predictor.py:29-71), from float64 numpy like the reference."""
    alphas = 1.0 - betas
    acp = np.cumprod(alphas, axis=0)
    acp_prev = np.append(1.0, acp[:-1])
    var = betas * (1.0 - acp_prev) / (1.0 - acp)
    return dict(sqrt_recip_alphas_cumprod=_f32(np.sqrt(1.0 / acp)), sqrt_recipm1_alphas_cumprod=_f32(np.sqrt(1.0 / acp - 1)),
                posterior_log_variance_clipped=_f32(np.log(np.maximum(var, 1e-20))),
                posterior_mean_coef1=_f32(betas * np.sqrt(acp_prev) / (1.0 - acp)),
                posterior_mean_coef2=_f32((1.0 - acp_prev) * np.sqrt(alphas) / (1.0 - acp)), clip_min=_f32(-1.0), clip_max=_f32(1.0))


def naive_table_from_buffers(buf: dict, chunks) -> np.ndarray:
    """Rows of the DDPM update rule from the predictor's BUFFERS -- in the reference they are `register_buffer`s, i.e. whatever the
    checkpoint holds is what sampling uses (noise_predictor.py:73-104): x0 = sr[t] x - srm1[t] eps, clamp(clip_min, clip_max),
    mean = c1[t] x0 + c2[t] x, + (t > 0) exp(0.5 logvar[t]) noise.  One-element fp32 tensor ops, as the reference evaluates them."""
    sr, srm1 = buf["sqrt_recip_alphas_cumprod"], buf["sqrt_recipm1_alphas_cumprod"]
    c1, c2, logvar = buf["posterior_mean_coef1"], buf["posterior_mean_coef2"], buf["posterior_log_variance_clipped"]
    lo, hi = float(buf["clip_min"]), float(buf["clip_max"])
    tab = np.zeros((len(chunks), FDX_ROW), np.float32)
    for r, t in enumerate(chunks):
        nonzero = torch.tensor(1.0 if t > 0 else 0.0)
        scale = nonzero * (0.5 * logvar[t]).exp()  # noise_predictor.py:102-104
        tab[r, :8] = [float(v) for v in (t, sr[t], srm1[t], c1[t], c2[t], scale, lo, hi)]
    return tab


def naive_table(betas: np.ndarray, chunks) -> np.ndarray:
    return naive_table_from_buffers(naive_buffers(betas), chunks)


# ------------------------------------------------------------------ PLMS
def plms_table_from_buffers(acp: torch.Tensor, chunks, interval: int) -> np.ndarray:
    """noise_predictor.py:118-131 from the predictor's `alphas_cumprod` BUFFER (fp32)."""
    tab = np.zeros((len(chunks), FDX_ROW), np.float32)
    for r, t in enumerate(chunks):
        tp = max(t - interval, 0)  # diffusion.py:280-281
        a_t, a_prev = acp[t], acp[tp]
        a_t_sq, a_prev_sq = a_t.sqrt(), a_prev.sqrt()
        A = a_prev - a_t
        P = 1 / (a_t_sq * (a_t_sq + a_prev_sq))
        Q = 1 / (a_t_sq * (((1 - a_prev) * a_t).sqrt() + ((1 - a_t) * a_prev).sqrt()))
        tab[r, :5] = [float(v) for v in (t, tp, A, P, Q)]
    return tab


def plms_table(betas: np.ndarray, chunks, interval: int) -> np.ndarray:
    return plms_table_from_buffers(_f32(np.cumprod(1.0 - betas, axis=0)), chunks, interval)


# ------------------------------------------------------------------ UniPC (bh2, order 2, multistep, time_uniform)
class _DiscreteVP:
    def __init__(self, betas: np.ndarray):
        b = torch.from_numpy(np.asarray(betas, np.float64))
        self.log_alpha = (0.5 * torch.log(1 - b).cumsum(dim=0)).to(torch.float32)  # uni_pc.py:84-99
        self.N = len(self.log_alpha)
        self.knots = torch.linspace(0.0, 1.0, self.N + 1)[1:].to(torch.float32)

    def log_alpha_t(self, t):  # piecewise linear, linear extrapolation (uni_pc.py:826-875)
        j = torch.clamp(torch.searchsorted(self.knots, t, right=False) - 1, 0, self.N - 2)
        x0, x1 = self.knots[j], self.knots[j + 1]
        y0, y1 = self.log_alpha[j], self.log_alpha[j + 1]
        return y0 + (t - x0) * (y1 - y0) / (x1 - x0)

    def sigma(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.log_alpha_t(t)))

    def lam(self, t):
        la = self.log_alpha_t(t)
        return la - 0.5 * torch.log(1.0 - torch.exp(2.0 * la))


def unipc_table(betas: np.ndarray, steps: int) -> np.ndarray:
    if steps < 2:
        raise AssertionError("UniPC needs steps >= order (2)")  # uni_pc.py:741
    ns = _DiscreteVP(betas)
    ts = torch.linspace(1.0, 1.0 / ns.N, steps + 1)
    tab = np.zeros((steps + 1, FDX_ROW), np.float32)

    def t_input(t):  # uni_pc.py:220-223
        return (t - 1.0 / ns.N) * ns.N

    t0 = ts[0].view(-1)
    tab[0, :3] = [float(v) for v in (t_input(t0), ns.sigma(t0), torch.exp(ns.log_alpha_t(t0)))]
    t_hist = [t0]
    for step in range(1, steps + 1):
        t = ts[step].view(-1)
        order = step if step < 2 else min(2, steps + 1 - step)
        use_corr = step < steps
        tp0 = t_hist[-1]
        lam0, lam_t = ns.lam(tp0), ns.lam(t)
        sig0, sig_t = ns.sigma(tp0), ns.sigma(t)
        alpha_t = torch.exp(ns.log_alpha_t(t))
        h = lam_t - lam0
        rks = []
        rk = torch.ones(1)
        if order == 2:
            rk = (ns.lam(t_hist[-2]) - lam0) / h
            rks.append(rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)  # bh2
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        rho0, rho1 = 0.0, 0.5  # order-1 corrector uses [0.5] (uni_pc.py:657-658)
        if use_corr and order == 2:
            rho = torch.linalg.solve(torch.stack(R), torch.cat(b))
            rho0, rho1 = rho[0], rho[1]
        tab[step, :11] = [float(v) for v in (t_input(t), sig_t, alpha_t, sig_t / sig0, alpha_t * h_phi_1, alpha_t * B_h,
                                             rk, order, 1.0 if use_corr else 0.0, rho0, rho1)]
        t_hist = (t_hist + [t])[-2:]
    return tab


@functools.lru_cache(maxsize=64)
def _cached(kind: str, noise_schedule: str, timesteps: int, max_beta: float, s: float, interval: int, skip: int):
    betas = make_betas(noise_schedule, timesteps, max_beta, s)
    if kind == "unipc":
        # noise_predictor.py:187 -- steps from the FULL schedule, skip_steps does not shorten it
        return unipc_table(betas, timesteps // interval)
    chunks = timestep_chunks(timesteps, skip, interval)
    return naive_table(betas, chunks) if kind == "naive" else plms_table(betas, chunks, interval)


def sampler_table(kind: str, *, noise_schedule="linear", timesteps=1000, max_beta=0.01, s=0.008, interval=10,
                  skip_steps=0) -> Tuple[int, np.ndarray]:
    kind = kind.lower()
    if kind not in KINDS:
        raise NotImplementedError(f"Unknown noise predictor: {kind}")
    tab = _cached(kind, noise_schedule, int(timesteps), float(max_beta), float(s), int(interval), int(skip_steps))
    return KINDS[kind], tab
