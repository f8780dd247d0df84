"""CPU oracle for the RefineGAN generator.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Functional restatement (folded state-dict in, waveform out) of
fish_diffusion/modules/vocoders/refinegan/generator.py:
  ``RefineGANGenerator.forward`` :437-478 (ctor geometry :314-423), ``ResBlock.forward`` :63-75,
  ``AdaIN.forward`` :104-107, ``ParallelResBlock.forward`` :147-152, ``CombToothGen.forward`` :174-194
and of the scalar glue of ``RefineGAN.spec2wav`` (refinegan.py:67-78).
Both template generators are restated: "comb" (the default; what configs/_base_/archs/hifi_svc_v2.py and
configs/vocoder_refinegan.py use) and "sine" (``SineGen`` without overtones, :197-310).

All random draws are explicit inputs, in the order the reference draws them:
  noises[0]        [B, 1, L]        comb-tooth noise (generator.py:191)
  noises[1 + n]    [B, C, L_stage]  the n-th AdaIN call: stage-major, then branch k in (3, 7, 11), then (pre, post) (:104-107)
Pinned against the real reference module by oracle/make_golden.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

CONFIG = dict(sampling_rate=44100, hop_length=256, downsample_rates=(2, 2, 8, 8), upsample_rates=(8, 8, 2, 2),
              leaky_relu_slope=0.2, num_mels=128, start_channels=16)  # configs/vocoder_refinegan.py generator defaults


def _pad(k, d=1):
    return int((k * d - d) / 2)


def param_shapes(cfg: dict):
    """Folded (no weight_g / weight_v) parameter list with the reference's names."""
    c = cfg["start_channels"]
    out = []
    if cfg.get("template_generator", "comb") == "sine":
        out += [("template_gen.merge.0.weight", (1, 1)), ("template_gen.merge.0.bias", (1,))]
    out += [("template_conv.weight", (c, 1, 7)), ("template_conv.bias", (c,))]
    for i, _ in enumerate(cfg["downsample_rates"]):
        n = 2 * c
        for j in range(3):
            out += [(f"downsample_blocks.{i}.1.convs1.{j}.weight", (n, c if j == 0 else n, 7)), (f"downsample_blocks.{i}.1.convs1.{j}.bias", (n,)),
                    (f"downsample_blocks.{i}.1.convs2.{j}.weight", (n, n, 7)), (f"downsample_blocks.{i}.1.convs2.{j}.bias", (n,))]
        c = n
    out += [("mel_conv.weight", (c, cfg["num_mels"], 7)), ("mel_conv.bias", (c,))]
    c *= 2
    sf0 = int(np.prod(cfg["upsample_rates"][1:]))
    out += [("source_conv.weight", (c, 1, 2 * sf0)), ("source_conv.bias", (c,))]
    for i, _ in enumerate(cfg["upsample_rates"]):
        n = c // 2
        p = f"upsample_conv_blocks.{i}."
        out += [(p + "input_conv.weight", (n, c + c // 4, 7)), (p + "input_conv.bias", (n,))]
        for b, k in enumerate((3, 7, 11)):
            out += [(p + f"blocks.{b}.0.weight", (n,))]
            for j in range(3):
                out += [(p + f"blocks.{b}.1.convs1.{j}.weight", (n, n, k)), (p + f"blocks.{b}.1.convs1.{j}.bias", (n,)),
                        (p + f"blocks.{b}.1.convs2.{j}.weight", (n, n, k)), (p + f"blocks.{b}.1.convs2.{j}.bias", (n,))]
            out += [(p + f"blocks.{b}.2.weight", (n,))]
        c = n
    out += [("output_conv.weight", (1, c, 7)), ("output_conv.bias", (1,))]
    return out


def seeded_state(seed: int, cfg: dict) -> SD:
    """Deterministic synthetic folded weights, fan-in scaled so that every stage carries O(1) signal (the reference's
    N(0, 0.01) init drives untrained activations to ~0 and would make parity vacuous); AdaIN weights ~ U(0.05, 0.2)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in param_shapes(cfg):
        if key.endswith("bias"):
            sd[key] = torch.randn(shape, generator=g) * 0.02
        elif key == "template_gen.merge.0.weight":
            sd[key] = 4.0 + 2.0 * torch.rand(shape, generator=g)      # O(1) template from a 0.1-amplitude sine
        elif len(shape) == 1:
            sd[key] = 0.05 + 0.15 * torch.rand(shape, generator=g)
        else:
            sd[key] = torch.randn(shape, generator=g) * (1.0 / (shape[1] * shape[2])) ** 0.5
    return sd


def n_noise_tensors(cfg: dict) -> int:
    return 1 + 6 * len(cfg["upsample_rates"])


def comb_tooth(f0_up: torch.Tensor, noise: torch.Tensor, sr: int, wave_amp=0.1, noise_std=0.003) -> torch.Tensor:
    """generator.py:174-194.  f0_up [B,1,L]."""
    x = torch.cumsum(f0_up / sr, axis=2)
    x = x - torch.round(x)
    comb = torch.sinc(sr * x / (f0_up + 1e-3)) * wave_amp
    uv = (f0_up > 0).float()
    noise_amp = uv * noise_std + (1 - uv) * wave_amp / 3
    return comb * uv + noise_amp * noise


def sine_template(sd: SD, f0_up: torch.Tensor, noise: torch.Tensor, sr: int, sine_amp=0.1, noise_std=0.003) -> torch.Tensor:
    """SineGen.forward with harmonic_num = 0 (generator.py:246-310).  f0_up [B,1,L]; noise [B,1,L] (the reference draws it as
    [B,L,1]: same values, transposed).  rand_ini is drawn and zeroed for the fundamental (:254-258): the initial phase is 0."""
    f0 = f0_up.transpose(1, 2)                                   # [B, L, 1]
    rad = (f0 / sr) % 1
    tmp = torch.cumsum(rad, 1) % 1
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = ((tmp[:, 1:, :] - tmp[:, :-1, :]) < 0) * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi)
    sines[f0 > sr // 2] = 0
    sine_waves = sines * sine_amp
    uv = (f0 > 0).to(f0.dtype)
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3
    sine_waves = sine_waves * uv + noise_amp * noise.transpose(1, 2)
    return torch.tanh(F.linear(sine_waves, sd["template_gen.merge.0.weight"], sd["template_gen.merge.0.bias"])).transpose(1, 2)


def _resblock(sd: SD, prefix: str, x, k: int, slope: float, same: bool):
    for j, d in enumerate((1, 3, 5)):
        xt = F.leaky_relu(x, slope)
        xt = F.conv1d(xt, sd[prefix + f"convs1.{j}.weight"], sd[prefix + f"convs1.{j}.bias"], dilation=d, padding=_pad(k, d))
        xt = F.leaky_relu(xt, slope)
        xt = F.conv1d(xt, sd[prefix + f"convs2.{j}.weight"], sd[prefix + f"convs2.{j}.bias"], dilation=d, padding=_pad(k, d))
        x = xt + x if (j != 0 or same) else xt
    return x


def generator_forward(sd: SD, cfg: dict, mel: torch.Tensor, f0: torch.Tensor, noises: List[torch.Tensor],
                      taps: Optional[dict] = None) -> torch.Tensor:
    """generator.py:437-478.  mel [B,num_mels,T]; f0 [B,1,T]; returns [B,1,T*hop]."""
    slope, sr, hop = cfg["leaky_relu_slope"], cfg["sampling_rate"], cfg["hop_length"]
    it = iter(noises)
    f0_up = F.interpolate(f0, size=mel.shape[-1] * hop, mode="linear")
    if cfg.get("template_generator", "comb") == "sine":
        template = sine_template(sd, f0_up, next(it), sr)
    else:
        template = comb_tooth(f0_up, next(it), sr)
    if taps is not None:
        taps["template"] = template
    x = F.conv1d(template, sd["template_conv.weight"], sd["template_conv.bias"], padding=3)
    downs = []
    for i, rate in enumerate(cfg["downsample_rates"]):
        x = F.leaky_relu(x, slope)
        downs.append(x)
        x = F.interpolate(x, scale_factor=1 / rate, mode="linear")
        x = _resblock(sd, f"downsample_blocks.{i}.1.", x, 7, slope, same=False)
    x = torch.cat([x, F.conv1d(mel, sd["mel_conv.weight"], sd["mel_conv.bias"], padding=3)], dim=1)
    if taps is not None:
        taps["bottleneck"] = x
    sf0 = int(np.prod(cfg["upsample_rates"][1:]))
    for i, (rate, down) in enumerate(zip(cfg["upsample_rates"], reversed(downs))):
        x = F.leaky_relu(x, slope)
        x = F.interpolate(x, scale_factor=float(rate), mode="linear")
        if i == 0:
            x = x + F.conv1d(template, sd["source_conv.weight"], sd["source_conv.bias"], stride=sf0, padding=sf0 // 2)
        x = torch.cat([x, down], dim=1)
        p = f"upsample_conv_blocks.{i}."
        x = F.conv1d(x, sd[p + "input_conv.weight"], sd[p + "input_conv.bias"], padding=3)
        results = []
        for b, k in enumerate((3, 7, 11)):
            y = F.leaky_relu(x + next(it) * sd[p + f"blocks.{b}.0.weight"][None, :, None], slope)
            y = _resblock(sd, p + f"blocks.{b}.1.", y, k, slope, same=True)
            y = F.leaky_relu(y + next(it) * sd[p + f"blocks.{b}.2.weight"][None, :, None], slope)
            results.append(y)
        x = torch.mean(torch.stack(results), dim=0)
        if taps is not None:
            taps[f"up_{i}"] = x
    x = F.leaky_relu(x, slope)
    x = F.conv1d(x, sd["output_conv.weight"], sd["output_conv.bias"], padding=3)
    return torch.tanh(x)


def noise_shapes(cfg: dict, B: int, T: int):
    """Shapes of the draws ``generator_forward`` consumes, in order."""
    L = T * cfg["hop_length"]
    shapes = [(B, 1, L)]
    c = cfg["start_channels"] * 2 ** len(cfg["downsample_rates"]) * 2
    length = T
    for rate in cfg["upsample_rates"]:
        c //= 2
        length *= rate
        shapes += [(B, c, length)] * 6
    return shapes


def spec2wav(sd: SD, cfg: dict, mel, f0, noises, *, key_shift=0, use_natural_log=True):
    """refinegan.py:67-78.  mel [num_mels,T], f0 [T] -> wav [T*hop]."""
    c = mel[None]
    f0 = f0 * 2 ** (key_shift / 12)
    if use_natural_log is False:
        c = 2.30259 * c
    return generator_forward(sd, cfg, c, f0[None, None].to(c.dtype), noises).view(-1)
