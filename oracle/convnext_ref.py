"""CPU oracle for the ConvNext denoiser.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Functional restatement (state-dict in, tensor out) of fish_diffusion/modules/convnext.py:
  ``ConvNeXtBlock.forward`` :56-92, ``CrossAttentionBlock.forward`` :127-152 and ``ConvNext.forward`` :211-262, registered as
  DENOISERS "ConvNextDenoiser" (archs/diffsinger/diffusions/builder.py:12).  Same call contract as the WaveNet denoiser.
Pinned against the real module by oracle/make_golden.py: bit-exact with and without ``cross_attention`` (since round 6 the
attention of a CrossAttentionBlock is evaluated in torch's own operation order, oracle/tfdec_ref.py::mha; the goldens hold the
REAL module's outputs).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .wavenet_ref import diffusion_embedding, layer_dilations

SD = Dict[str, torch.Tensor]


def cross_shapes(p, dim, h):
    """A CrossAttentionBlock's tensors (convnext.py:95-112) in its state_dict order."""
    out = [(p + "position_scale_query", (1,)), (p + "position_scale_key", (1,)), (p + "positional_embedding", (4096, dim))]
    for att in ("self_attn", "multihead_attn"):
        out += [(p + att + ".in_proj_weight", (3 * dim, dim)), (p + att + ".in_proj_bias", (3 * dim,)),
                (p + att + ".out_proj.weight", (dim, dim)), (p + att + ".out_proj.bias", (dim,))]
    out += [(p + "linear1.weight", (h, dim)), (p + "linear1.bias", (h,)), (p + "linear2.weight", (dim, h)), (p + "linear2.bias", (dim,))]
    for n in ("norm1", "norm2", "norm3"):
        out += [(p + n + ".weight", (dim,)), (p + n + ".bias", (dim,))]
    out += [(p + "diffusion_step_projection.weight", (dim, dim, 1)), (p + "diffusion_step_projection.bias", (dim,))]
    return out


def layer_plan(num_layers, cross_every=0):
    """The reference's mixed `residual_layers` list (convnext.py:186-201): [(kind, module index, conv-layer index)]."""
    plan, j = [], 0
    for i in range(num_layers):
        if cross_every and i % cross_every == 0:
            plan.append(("cross", j, i))
            j += 1
        plan.append(("conv", j, i))
        j += 1
    return plan


def param_shapes(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=20, cross_every=0):
    h = dim * mlp_factor
    out = [("input_projection.weight", (dim, mel_channels, 1)), ("input_projection.bias", (dim,)),
           ("diffusion_embedding.1.weight", (h, dim)), ("diffusion_embedding.1.bias", (h,)),
           ("diffusion_embedding.3.weight", (dim, h)), ("diffusion_embedding.3.bias", (dim,)),
           ("conditioner_projection.0.weight", (h, condition_dim, 1)), ("conditioner_projection.0.bias", (h,)),
           ("conditioner_projection.2.weight", (dim, h, 1)), ("conditioner_projection.2.bias", (dim,))]
    for kind, j, _ in layer_plan(num_layers, cross_every):
        p = f"residual_layers.{j}."
        if kind == "cross":
            out += cross_shapes(p, dim, h)
            continue
        out += [(p + "gamma", (dim,)), (p + "dwconv.weight", (dim, 1, 7)), (p + "dwconv.bias", (dim,)),
                (p + "norm.weight", (dim,)), (p + "norm.bias", (dim,)),
                (p + "pwconv1.weight", (h, dim)), (p + "pwconv1.bias", (h,)),
                (p + "pwconv2.weight", (dim, h)), (p + "pwconv2.bias", (dim,)),
                (p + "diffusion_step_projection.weight", (dim, dim, 1)), (p + "diffusion_step_projection.bias", (dim,)),
                (p + "condition_projection.weight", (dim, dim, 1)), (p + "condition_projection.bias", (dim,))]
    out += [("output_projection.0.weight", (dim, dim, 1)), ("output_projection.0.bias", (dim,)),
            ("output_projection.2.weight", (mel_channels, dim, 1)), ("output_projection.2.bias", (mel_channels,))]
    return out


def seeded_state(seed: int, **cfg) -> SD:
    """Fan-in scaled synthetic weights; gamma ~ U(0.3, 1) (the reference's 1e-6 layer-scale init would switch every block off)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in param_shapes(**cfg):
        if key.endswith("positional_embedding"):
            from .tfdec_ref import positional_embedding
            sd[key] = positional_embedding(shape[1])
        elif "position_scale" in key:
            sd[key] = 0.5 + torch.rand(shape, generator=g)
        elif key.endswith("gamma"):
            sd[key] = 0.3 + 0.7 * torch.rand(shape, generator=g)
        elif key.endswith(("norm.weight", "norm1.weight", "norm2.weight", "norm3.weight")):
            sd[key] = 0.5 + torch.rand(shape, generator=g)
        elif key.endswith("bias"):
            sd[key] = torch.randn(shape, generator=g) * 0.05
        else:
            fan_in = shape[1] * (shape[2] if len(shape) == 3 else 1)
            sd[key] = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
    return sd


def cross_block(sd: SD, j: int, x, condition, step, x_masks=None, cond_masks=None):
    """convnext.py:127-152: x [B, D, T], condition [B, D, T] (already masked by the caller, :247-248), step [B, D, 1]."""
    from .tfdec_ref import mha
    p = f"residual_layers.{j}."
    D = x.shape[1]
    x = x + F.conv1d(step, sd[p + "diffusion_step_projection.weight"], sd[p + "diffusion_step_projection.bias"])
    x = x.transpose(1, 2)
    mem = condition.transpose(1, 2)
    x = x + sd[p + "positional_embedding"][: x.size(1)][None] * sd[p + "position_scale_query"]
    mem = mem + sd[p + "positional_embedding"][: mem.size(1)][None] * sd[p + "position_scale_key"]
    ln = lambda t, n: F.layer_norm(t, (D,), sd[p + n + ".weight"], sd[p + n + ".bias"], eps=1e-5)   # noqa: E731
    x = ln(x + mha(sd, p + "self_attn.", x, x, x_masks), "norm1")
    x = ln(x + mha(sd, p + "multihead_attn.", x, mem, cond_masks), "norm2")
    ff = F.linear(F.gelu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return ln(x + ff, "norm3").transpose(1, 2)


def block(sd: SD, i: int, x, condition, step, dilation: int, x_masks=None, cond_masks=None):
    """convnext.py:56-92.  condition None (cross-attention variant, :246-250): no condition term."""
    p = f"residual_layers.{i}."
    residual = x
    x = x + F.conv1d(step, sd[p + "diffusion_step_projection.weight"], sd[p + "diffusion_step_projection.bias"])
    if condition is not None:
        if cond_masks is not None:
            condition = condition.masked_fill(cond_masks[:, None, :], 0.0)
        x = x + F.conv1d(condition, sd[p + "condition_projection.weight"], sd[p + "condition_projection.bias"])
    if x_masks is not None:
        x = x.masked_fill(x_masks[:, None, :], 0.0)
    x = F.conv1d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], groups=x.shape[1], dilation=dilation, padding=int(dilation * 6 / 2))
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-6)
    x = F.linear(x, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])
    x = F.gelu(x)
    x = F.linear(x, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
    x = sd[p + "gamma"] * x
    x = residual + x.transpose(1, 2)
    if x_masks is not None:
        x = x.masked_fill(x_masks[:, None, :], 0.0)
    return x


def convnext_forward(sd: SD, x, diffusion_step, conditioner, x_masks=None, cond_masks=None, *, num_layers=20, dilation_cycle=4,
                     cross_every=0):
    """convnext.py:211-262.  cross_every = cross_every_n_layers when cross_attention else 0."""
    use_4_dim = x.dim() == 4
    if use_4_dim:
        x = x[:, 0]
    assert x.dim() == 3, f"mel must be 3 dim tensor, but got {x.dim()}"
    dim = sd["input_projection.weight"].shape[0]
    x = F.gelu(F.conv1d(x, sd["input_projection.weight"], sd["input_projection.bias"]))
    e = diffusion_embedding(diffusion_step, dim)
    e = F.linear(F.gelu(F.linear(e, sd["diffusion_embedding.1.weight"], sd["diffusion_embedding.1.bias"])),
                 sd["diffusion_embedding.3.weight"], sd["diffusion_embedding.3.bias"])
    step = e.unsqueeze(-1)
    condition = F.conv1d(F.gelu(F.conv1d(conditioner, sd["conditioner_projection.0.weight"], sd["conditioner_projection.0.bias"])),
                         sd["conditioner_projection.2.weight"], sd["conditioner_projection.2.bias"])
    if x_masks is not None:
        x = x.masked_fill(x_masks[:, None, :], 0.0)
    if cond_masks is not None:
        condition = condition.masked_fill(cond_masks[:, None, :], 0.0)
    dil = layer_dilations(num_layers, dilation_cycle)
    for kind, j, i in layer_plan(num_layers, cross_every):
        if kind == "cross":
            x = cross_block(sd, j, x, condition, step, x_masks, cond_masks)
        else:
            x = block(sd, j, x, None if cross_every else condition, step, dil[i], x_masks, cond_masks)
    x = F.conv1d(F.gelu(F.conv1d(x, sd["output_projection.0.weight"], sd["output_projection.0.bias"])),
                 sd["output_projection.2.weight"], sd["output_projection.2.bias"])
    if x_masks is not None:
        x = x.masked_fill(x_masks[:, None, :], 0.0)
    return x[:, None] if use_4_dim else x
