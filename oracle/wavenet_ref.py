"""CPU oracle for the WaveNet-residual denoiser.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

A functional (state-dict in, tensor out) restatement of
fish_diffusion/modules/wavenet.py -- ``WaveNet.forward`` :194-236, ``ResidualBlock.forward`` :106-120,
``DiffusionEmbedding.forward`` :20-27, ``Mish`` :8-10 -- in plain torch fp32 on CPU.  It is pinned
against the real reference module by oracle/make_golden.py (fixtures in tests/golden/).

State-dict keys are the reference's own (wavenet.py:35,66,88-104,168-191):
  input_projection.conv.{weight,bias}, mlp.{0,2}.linear.{weight[,bias]},
  residual_layers.{i}.{conv_layer.conv,conditioner_projection.conv,output_projection.conv}.{weight,bias},
  residual_layers.{i}.diffusion_projection.linear.{weight[,bias]},
  skip_projection.conv.{weight,bias}, output_projection.conv.{weight,bias}
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def wavenet_param_shapes(mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=20,
                         use_linear_bias=False):
    """(key, shape) list in the reference's registration order."""
    C = residual_channels
    out = [("input_projection.conv.weight", (C, mel_channels, 1)), ("input_projection.conv.bias", (C,))]
    out.append(("mlp.0.linear.weight", (4 * C, C)))
    if use_linear_bias:
        out.append(("mlp.0.linear.bias", (4 * C,)))
    out.append(("mlp.2.linear.weight", (C, 4 * C)))
    if use_linear_bias:
        out.append(("mlp.2.linear.bias", (C,)))
    for i in range(residual_layers):
        p = f"residual_layers.{i}."
        out += [(p + "conv_layer.conv.weight", (2 * C, C, 3)), (p + "conv_layer.conv.bias", (2 * C,))]
        out.append((p + "diffusion_projection.linear.weight", (C, C)))
        if use_linear_bias:
            out.append((p + "diffusion_projection.linear.bias", (C,)))
        out += [(p + "conditioner_projection.conv.weight", (2 * C, d_encoder, 1)),
                (p + "conditioner_projection.conv.bias", (2 * C,)),
                (p + "output_projection.conv.weight", (2 * C, C, 1)),
                (p + "output_projection.conv.bias", (2 * C,))]
    out += [("skip_projection.conv.weight", (C, C, 1)), ("skip_projection.conv.bias", (C,)),
            ("output_projection.conv.weight", (mel_channels, C, 1)), ("output_projection.conv.bias", (mel_channels,))]
    return out


def layer_dilations(residual_layers: int, dilation_cycle: Optional[int]):
    """wavenet.py:181 -- 2 ** (i % dilation_cycle) if dilation_cycle else 1."""
    return [2 ** (i % dilation_cycle) if dilation_cycle else 1 for i in range(residual_layers)]


def diffusion_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """wavenet.py:20-27.  t [B] (long or float) -> [B, dim]."""
    half = dim // 2
    scale = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, device=t.device) * -scale)
    arg = t[:, None] * freq[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def mish(x):
    """wavenet.py:8-10."""
    return x * torch.tanh(F.softplus(x))


def step_mlp(sd: SD, t: torch.Tensor, C: int) -> torch.Tensor:
    """wavenet.py:214-215 -> [B, C]."""
    e = diffusion_embedding(t, C)
    e = F.linear(e, sd["mlp.0.linear.weight"], sd.get("mlp.0.linear.bias"))
    e = mish(e)
    return F.linear(e, sd["mlp.2.linear.weight"], sd.get("mlp.2.linear.bias"))


def residual_block(sd: SD, i: int, x, cond, step, dilation: int, operand_round=None):
    """wavenet.py:106-120.  (`operand_round`: see bf16_storage_model -- None for the reference arithmetic.)"""
    p = f"residual_layers.{i}."
    s = F.linear(step, sd[p + "diffusion_projection.linear.weight"],
                 sd.get(p + "diffusion_projection.linear.bias")).unsqueeze(-1)
    c = F.conv1d(cond, sd[p + "conditioner_projection.conv.weight"], sd[p + "conditioner_projection.conv.bias"])
    y = x + s
    if operand_round is not None:
        y = operand_round(y)
    y = F.conv1d(y, sd[p + "conv_layer.conv.weight"], sd[p + "conv_layer.conv.bias"],
                 padding=dilation, dilation=dilation) + c
    gate, filt = torch.chunk(y, 2, dim=1)
    y = torch.sigmoid(gate) * torch.tanh(filt)
    if operand_round is not None:
        y = operand_round(y)
    y = F.conv1d(y, sd[p + "output_projection.conv.weight"], sd[p + "output_projection.conv.bias"])
    residual, skip = torch.chunk(y, 2, dim=1)
    return (x + residual) / math.sqrt(2.0), skip


def bf16_storage_model(sd: SD) -> "tuple[SD, callable]":
    """NOT a reference function: a CPU model of the library's opt-in bf16 storage mode (BASELINE configs[4] "bf16 storage,
    fp32 accumulate"), used only to tell rounding policy from layout bugs.  Returns (state with the two residual-block GEMM
    weights rounded to bf16, rounding function for their activation operands)."""
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    sd2 = dict(sd)
    for k in sd:
        if k.startswith("residual_layers.") and (k.endswith("conv_layer.conv.weight") or k.endswith("output_projection.conv.weight")):
            sd2[k] = rb(sd[k])
    return sd2, rb


def wavenet_forward(sd: SD, x, diffusion_step, conditioner, x_masks=None, cond_masks=None, *,
                    residual_layers=20, dilation_cycle=None, taps=None, operand_round=None):
    """wavenet.py:194-236.  x [B,M,T] (or [B,1,M,T]); diffusion_step [B]; conditioner [B,E,T].

    ``taps``: optional dict that receives intermediate activations (for per-layer parity tests).
    """
    use_4_dim = x.dim() == 4
    if use_4_dim:
        x = x[:, 0]
    assert x.dim() == 3, f"mel must be 3 dim tensor, but got {x.dim()}"
    C = sd["input_projection.conv.weight"].shape[0]
    x = F.relu(F.conv1d(x, sd["input_projection.conv.weight"], sd["input_projection.conv.bias"]))
    step = step_mlp(sd, diffusion_step, C)
    if x_masks is not None:
        x = x.masked_fill(x_masks[:, None], 0.0)
    if cond_masks is not None:
        conditioner = conditioner.masked_fill(cond_masks[:, None], 0.0)
    if taps is not None:
        taps["step"] = step
        taps["x_in"] = x
    skips = []
    for i, d in enumerate(layer_dilations(residual_layers, dilation_cycle)):
        x, s = residual_block(sd, i, x, conditioner, step, d, operand_round)
        skips.append(s)
        if taps is not None:
            taps[f"x_{i}"] = x
    x = torch.sum(torch.stack(skips), dim=0) / math.sqrt(residual_layers)
    if taps is not None:
        taps["skip_sum"] = x
    x = F.relu(F.conv1d(x, sd["skip_projection.conv.weight"], sd["skip_projection.conv.bias"]))
    x = F.conv1d(x, sd["output_projection.conv.weight"], sd["output_projection.conv.bias"])
    if x_masks is not None:
        x = x.masked_fill(x_masks[:, None], 0.0)
    return x[:, None] if use_4_dim else x


def seeded_wavenet_state(seed: int, mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=20,
                         use_linear_bias=True, out_std=0.02) -> SD:
    """Deterministic synthetic weights with the reference's initialisers' *statistics*
    (kaiming_normal_ for convs wavenet.py:75, xavier_uniform_ for linears :37; Conv1d default bias
    U(-1/sqrt(fan_in), ..)); the final output_projection, zero-initialised at wavenet.py:192, is
    redrawn N(0, out_std) because an all-zero denoiser output would make parity vacuous.
    Same torch version on both boxes => same tensors; tests also check a stored checksum."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in wavenet_param_shapes(mel_channels, d_encoder, residual_channels, residual_layers, use_linear_bias):
        if key.endswith("conv.weight"):
            fan_in = shape[1] * shape[2]
            std = math.sqrt(2.0 / fan_in)
            if key == "output_projection.conv.weight":
                std = out_std
            sd[key] = torch.randn(shape, generator=g) * std
        elif key.endswith("linear.weight"):
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif key.endswith("conv.bias"):
            wshape = dict(wavenet_param_shapes(mel_channels, d_encoder, residual_channels, residual_layers,
                                               use_linear_bias))[key[:-4] + "weight"]
            bound = 1.0 / math.sqrt(wshape[1] * wshape[2])
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:  # linear bias: reference sets 0.0 (wavenet.py:39); use small noise so the bias path is exercised
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
    return sd
