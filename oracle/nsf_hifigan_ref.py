"""CPU oracle for the NSF-HiFiGAN generator.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Functional restatement (state-dict in, waveform out; weight-norm already folded) of
fish_diffusion/modules/vocoders/nsf_hifigan/models.py:
  ``Generator.forward`` :407-438 (ctor geometry :354-405), ``ResBlock1.forward`` :103-110,
  ``ResBlock2.forward`` :150-155, ``SineGen`` :195-294, ``SourceModuleHnNSF.forward`` :337-350
and of the scalar glue in ``NsfHifiGAN.spec2wav`` (nsf_hifigan.py:72-85).

All random draws are explicit inputs:
  rand_ini  [B, 9]     uniform [0,1) with column 0 forced to 0 (models.py:210-214)
  src_noise [B, L, 9]  standard normal (models.py:289)
(the second randn at models.py:349 is discarded by the caller, :415).
Pinned against the real reference module by oracle/make_golden.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

CONFIG_V1 = dict(  # tools/nsf_hifigan/config_v1.json (hop 512)
    resblock="1", upsample_rates=[8, 8, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 2, 2],
    upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
    resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    num_mels=128, n_fft=2048, hop_size=512, win_size=2048, sampling_rate=44100, fmin=40, fmax=16000)
CONFIG_V1_256 = dict(CONFIG_V1, upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4],
                     hop_size=256)  # tools/nsf_hifigan/config_v1_256.json


def generator_param_shapes(h: dict):
    """Folded (no weight_g/weight_v) parameter list; names as models.py:362-403."""
    C0 = h["upsample_initial_channel"]
    rates, ksz = h["upsample_rates"], h["upsample_kernel_sizes"]
    out = [("m_source.l_linear.weight", (1, 9)), ("m_source.l_linear.bias", (1,)),
           ("conv_pre.weight", (C0, h["num_mels"], 7)), ("conv_pre.bias", (C0,))]
    n_res = 0
    for i, (u, k) in enumerate(zip(rates, ksz)):
        cin, cout = C0 // (2 ** i), C0 // (2 ** (i + 1))
        out += [(f"ups.{i}.weight", (cin, cout, k)), (f"ups.{i}.bias", (cout,))]
        if i + 1 < len(rates):
            s = int(np.prod(rates[i + 1:]))
            out += [(f"noise_convs.{i}.weight", (cout, 1, 2 * s)), (f"noise_convs.{i}.bias", (cout,))]
        else:
            out += [(f"noise_convs.{i}.weight", (cout, 1, 1)), (f"noise_convs.{i}.bias", (cout,))]
    for i in range(len(rates)):
        ch = C0 // (2 ** (i + 1))
        for k, dil in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
            if h["resblock"] == "1":
                for j in range(len(dil)):
                    out += [(f"resblocks.{n_res}.convs1.{j}.weight", (ch, ch, k)), (f"resblocks.{n_res}.convs1.{j}.bias", (ch,)),
                            (f"resblocks.{n_res}.convs2.{j}.weight", (ch, ch, k)), (f"resblocks.{n_res}.convs2.{j}.bias", (ch,))]
            else:
                for j in range(2):
                    out += [(f"resblocks.{n_res}.convs.{j}.weight", (ch, ch, k)), (f"resblocks.{n_res}.convs.{j}.bias", (ch,))]
            n_res += 1
    out += [("conv_post.weight", (1, ch, 7)), ("conv_post.bias", (1,))]
    return out


def seeded_generator_state(seed: int, h: dict, gain: float = 1.0) -> SD:
    """Deterministic synthetic folded weights.  The reference initialises convs N(0, 0.01)
    (models.py:17-20) which, untrained, drives every activation to ~0; we draw
    N(0, gain*sqrt(1/fan_in)) instead so that all stages carry O(1) signal and parity is meaningful."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in generator_param_shapes(h):
        if key.endswith("weight"):
            if key.startswith("ups."):
                fan_in = shape[0] * shape[2] / h["upsample_rates"][int(key.split(".")[1])]
            elif key.startswith("m_source"):
                fan_in = 9.0
            else:
                fan_in = shape[1] * shape[2]
            sd[key] = torch.randn(shape, generator=g) * (gain / np.sqrt(fan_in))
        else:
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
    return sd


# ------------------------------------------------------------------ source module
def sine_source(f0_up: torch.Tensor, rand_ini: torch.Tensor, src_noise: torch.Tensor, sd: SD,
                sampling_rate=44100, harmonic_num=8, sine_amp=0.1, noise_std=0.003, voiced_threshold=0.0):
    """models.py:195-294 + :337-350.  f0_up [B, L, 1] -> harmonic source [B, L, 1]."""
    dim = harmonic_num + 1
    mult = torch.arange(1, dim + 1, dtype=f0_up.dtype)
    f0_buf = f0_up * mult  # :270-275  f0 * (idx + 2) for the overtones, f0 for the fundamental
    rad = (f0_buf / sampling_rate) % 1  # :203
    rad = rad.clone()
    rad[:, 0, :] = rad[:, 0, :] + rand_ini  # :214
    tmp = torch.cumsum(rad, 1) % 1  # :224
    over = (tmp[:, 1:, :] - tmp[:, :-1, :]) < 0  # :225
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over * -1.0  # :226-227
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi)  # :229-231
    sine_waves = sines * sine_amp  # :278
    uv = torch.ones_like(f0_up) * (f0_up > voiced_threshold)  # :189-193
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3  # :288
    noise = noise_amp * src_noise  # :289
    sine_waves = sine_waves * uv + noise  # :293
    return torch.tanh(F.linear(sine_waves, sd["m_source.l_linear.weight"], sd["m_source.l_linear.bias"]))  # :346


# ------------------------------------------------------------------ generator
def _resblock1(sd: SD, n: int, x, k: int, dils):
    """models.py:103-110."""
    for j, d in enumerate(dils):
        xt = F.leaky_relu(x, 0.1)
        xt = F.conv1d(xt, sd[f"resblocks.{n}.convs1.{j}.weight"], sd[f"resblocks.{n}.convs1.{j}.bias"],
                      dilation=d, padding=int((k * d - d) / 2))
        xt = F.leaky_relu(xt, 0.1)
        xt = F.conv1d(xt, sd[f"resblocks.{n}.convs2.{j}.weight"], sd[f"resblocks.{n}.convs2.{j}.bias"],
                      dilation=1, padding=int((k - 1) / 2))
        x = xt + x
    return x


def _resblock2(sd: SD, n: int, x, k: int, dils):
    """models.py:150-155.  The reference's ``F.leaky_relu(x, LRELU_SLOPE, inplace=True)`` (:152) rewrites ``x`` itself, so (a) the residual
    ``xt + x`` (:154) adds the ACTIVATED x, and (b) the first iteration rewrites the CALLER's tensor: in ``Generator.forward`` (:426-431)
    the same ``x`` is handed to the next ResBlock2 of the stage, which therefore starts from ``leaky_relu(x)``, the third from
    ``leaky_relu(leaky_relu(x))``.  Restated with the same in-place op so the caller sees it too (pinned: make_golden round5)."""
    for j, d in enumerate(dils):
        xt = F.leaky_relu(x, 0.1, inplace=True)          # xt IS x from here on
        xt = F.conv1d(xt, sd[f"resblocks.{n}.convs.{j}.weight"], sd[f"resblocks.{n}.convs.{j}.bias"],
                      dilation=d, padding=int((k * d - d) / 2))
        x = xt + x
    return x


def generator_forward(sd: SD, h: dict, mel: torch.Tensor, f0: torch.Tensor, rand_ini: torch.Tensor,
                      src_noise: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """models.py:407-438.  mel [B, num_mels, T] (natural-log scale), f0 [B, T] -> wav [B, 1, T*hop]."""
    if f0.ndim == 2:
        f0 = f0[:, None]
    L = mel.shape[-1] * h["hop_size"]
    f0_up = F.interpolate(f0, size=L, mode="linear").transpose(1, 2)  # :411-413
    har = sine_source(f0_up, rand_ini, src_noise, sd, sampling_rate=h["sampling_rate"]).transpose(1, 2)
    if taps is not None:
        taps["har_source"] = har
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    rates, ksz = h["upsample_rates"], h["upsample_kernel_sizes"]
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if i + 1 < len(rates):
            s = int(np.prod(rates[i + 1:]))
            xs_src = F.conv1d(har, sd[f"noise_convs.{i}.weight"], sd[f"noise_convs.{i}.bias"], stride=s, padding=s // 2)
        else:
            xs_src = F.conv1d(har, sd[f"noise_convs.{i}.weight"], sd[f"noise_convs.{i}.bias"])
        x = x + xs_src
        xs = None
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            rb = _resblock1 if h["resblock"] == "1" else _resblock2
            r = rb(sd, i * nk + j, x, rk, rd)
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps[f"stage_{i}"] = x
    x = F.leaky_relu(x)  # default slope 0.01, models.py:434
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def spec2wav(sd: SD, h: dict, mel: torch.Tensor, f0: torch.Tensor, rand_ini, src_noise, *,
             key_shift=0, use_natural_log=True) -> torch.Tensor:
    """nsf_hifigan.py:72-85.  mel [num_mels, T], f0 [T] -> wav [T*hop]."""
    c = mel[None]
    if key_shift is not None and key_shift != 0:
        f0 = f0 * 2 ** (key_shift / 12)
    if use_natural_log is False:
        c = 2.30259 * c
    return generator_forward(sd, h, c, f0[None].to(c.dtype), rand_ini, src_noise).view(-1)


def fold_weight_norm(state: SD) -> SD:
    """torch.nn.utils.remove_weight_norm semantics (dim=0): w = g * v / ||v||, norm over all dims but 0.
    Used to accept reference checkpoints whose keys are in weight_g / weight_v form (nsf_hifigan.py:38-52)."""
    out = {}
    for k, v in state.items():
        if k.endswith("weight_v"):
            g = state[k[:-1] + "g"]
            norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
            out[k[:-2]] = v * (g / norm)
        elif k.endswith("weight_g"):
            continue
        else:
            out[k] = v
    return out
