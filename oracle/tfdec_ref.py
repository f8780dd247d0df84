"""CPU oracle for the TransformerDecoderDenoiser.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Functional restatement (state-dict in, tensor out) of fish_diffusion/modules/convnext.py:263-379 (`TransformerDecoderDenoiser`),
registered as DENOISERS "TransformerDecoderDenoiser" (archs/diffsinger/diffusions/builder.py:13).  Its layers are
`torch.nn.TransformerDecoderLayer(d_model, nhead=8, dim_feedforward, activation="gelu", batch_first=True)` -- un-vendored torch
(pinned torch==2.0.1 in the reference's pdm.lock); restated here from its documented post-norm algorithm:
    x = norm1(x + self_attn(x, x, x, key_padding_mask=tgt_kpm));  x = norm2(x + mha(x, mem, mem, key_padding_mask=mem_kpm));
    x = norm3(x + linear2(gelu(linear1(x))))          (eval mode: every dropout is the identity)
with multi-head attention = softmax(q k^T / sqrt(d_head) + (-inf at masked keys)) v per head, packed in_proj.
Pinned against the real module (real nn.TransformerDecoderLayer instances) by oracle/make_golden.py -- BIT FOR BIT since round 6
(rounds 1-5: 3e-6 abs): `mha` below evaluates the formula above in the operation order torch's MultiheadAttention itself takes
(matmul-then-bias projections, pre-scaled q, the fused masked softmax, sdpa for cross-attention), forwards and 50-100-step sampler
runs alike.  The golden fixtures hold the REAL module's outputs.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .wavenet_ref import diffusion_embedding

SD = Dict[str, torch.Tensor]
NHEAD = 8


def positional_embedding(dim: int, n: int = 4096) -> torch.Tensor:
    """convnext.py:317-329."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float) * -emb)
    emb = torch.arange(n, dtype=torch.float).unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1).view(n, -1)


def param_shapes(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=12):
    h = dim * mlp_factor
    out = [("position_scale_query", (1,)), ("position_scale_key", (1,)), ("positional_embedding", (4096, dim)),
           ("input_projection.0.weight", (h, mel_channels, 1)), ("input_projection.0.bias", (h,)),
           ("input_projection.2.weight", (dim, h, 1)), ("input_projection.2.bias", (dim,)),
           ("diffusion_embedding.1.weight", (h, dim)), ("diffusion_embedding.1.bias", (h,)),
           ("diffusion_embedding.3.weight", (dim, h)), ("diffusion_embedding.3.bias", (dim,)),
           ("condition_projection.0.weight", (h, condition_dim, 1)), ("condition_projection.0.bias", (h,)),
           ("condition_projection.2.weight", (dim, h, 1)), ("condition_projection.2.bias", (dim,))]
    for i in range(num_layers):
        p = f"layers.{i}."
        out += [(p + "self_attn.in_proj_weight", (3 * dim, dim)), (p + "self_attn.in_proj_bias", (3 * dim,)),
                (p + "self_attn.out_proj.weight", (dim, dim)), (p + "self_attn.out_proj.bias", (dim,)),
                (p + "multihead_attn.in_proj_weight", (3 * dim, dim)), (p + "multihead_attn.in_proj_bias", (3 * dim,)),
                (p + "multihead_attn.out_proj.weight", (dim, dim)), (p + "multihead_attn.out_proj.bias", (dim,)),
                (p + "linear1.weight", (h, dim)), (p + "linear1.bias", (h,)),
                (p + "linear2.weight", (dim, h)), (p + "linear2.bias", (dim,)),
                (p + "norm1.weight", (dim,)), (p + "norm1.bias", (dim,)), (p + "norm2.weight", (dim,)), (p + "norm2.bias", (dim,)),
                (p + "norm3.weight", (dim,)), (p + "norm3.bias", (dim,))]
    out += [("output_projection.0.weight", (dim, dim, 1)), ("output_projection.0.bias", (dim,)),
            ("output_projection.2.weight", (mel_channels, dim, 1)), ("output_projection.2.bias", (mel_channels,))]
    return out


def seeded_state(seed: int, **cfg) -> SD:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in param_shapes(**cfg):
        if key == "positional_embedding":
            sd[key] = positional_embedding(shape[1])
        elif key.startswith("position_scale"):
            sd[key] = 0.5 + torch.rand(shape, generator=g)
        elif "norm" in key and key.endswith("weight"):
            sd[key] = 0.5 + torch.rand(shape, generator=g)
        elif key.endswith("bias"):
            sd[key] = torch.randn(shape, generator=g) * 0.05
        else:
            fan_in = shape[1] * (shape[2] if len(shape) == 3 else 1)
            sd[key] = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
    return sd


def mha(sd: SD, p: str, q_in, kv_in, key_padding_mask: Optional[torch.Tensor]):
    """nn.MultiheadAttention(batch_first=True, packed in_proj), eval, no grad: q_in [B, Tq, D], kv_in [B, Tk, D].

    Restated operation by operation in the form torch itself evaluates it (torch/nn/modules/activation.py MultiheadAttention.forward and
    torch/nn/functional.py multi_head_attention_forward; the golden run is under no_grad), because fp32 results depend on it:
      * self-attention (query is key is value) takes the native fast path (`torch._native_multi_head_attention`): ONE packed projection as
        matmul-then-bias, q scaled by 1 / sqrt(d_head) BEFORE the product, softmax over the keys -- through `torch._masked_softmax`
        (mask_type 1 = key padding) when a padding mask is given: that kernel sums a row sequentially in double, `softmax(masked_fill(-inf))`
        sums in vector lanes -- then attn @ v and the out-projection as addmm;
      * cross-attention goes through `F.multi_head_attention_forward`: inputs transposed to [T, B, D] views, q and the packed (k, v)
        projections again matmul-then-bias (what `F.linear` does for a non-contiguous 3-D input and a parameter), the padding mask as an
        additive float mask into `F.scaled_dot_product_attention`, out-projection as addmm on [Tq * B, D].
    (`addmm` and matmul-then-bias round differently; so do the fused and the plain softmax.)  In this form the restatement equals the real
    module BIT FOR BIT on the torch build the fixtures were made with -- oracle/make_golden.py asserts it, tests/test_oracle_golden.py re-checks
    equality on that build and 1e-5 rel on any other."""
    D = q_in.shape[-1]
    W, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    B, Tq, _ = q_in.shape
    Tk = kv_in.shape[1]
    dh = D // NHEAD
    if q_in is kv_in:
        qkv = (q_in.reshape(B * Tq, D) @ W.t()).view(B, Tq, 3 * D) + b
        q, k, v = (c.reshape(B, Tq, NHEAD, dh).transpose(1, 2) for c in qkv.chunk(3, dim=-1))
        s = (q * (1.0 / math.sqrt(dh))) @ k.transpose(-1, -2)
        a = torch.softmax(s, dim=-1) if key_padding_mask is None else torch._masked_softmax(s, key_padding_mask, 3, 1)
        o = (a @ v).transpose(1, 2).reshape(B, Tq, D)
        return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])
    qt, kt = q_in.transpose(1, 0), kv_in.transpose(1, 0)
    q = (qt.reshape(Tq * B, D) @ W[:D].t()).view(Tq, B, D) + b[:D]
    kv = (kt.reshape(Tk * B, D) @ W[D:].t()).view(Tk, B, 2 * D) + b[D:]
    kv = kv.unflatten(-1, (2, D)).unsqueeze(0).transpose(0, -2).squeeze(-2).contiguous()
    q = q.view(Tq, B * NHEAD, dh).transpose(0, 1).view(B, NHEAD, Tq, dh)
    k = kv[0].view(Tk, B * NHEAD, dh).transpose(0, 1).view(B, NHEAD, Tk, dh)
    v = kv[1].view(Tk, B * NHEAD, dh).transpose(0, 1).view(B, NHEAD, Tk, dh)
    mask = None
    if key_padding_mask is not None:
        mask = torch.zeros(B, 1, 1, Tk, dtype=q.dtype).masked_fill(key_padding_mask[:, None, None, :], float("-inf")).expand(-1, NHEAD, -1, -1)
    o = F.scaled_dot_product_attention(q, k, v, mask, 0.0, False)
    o = o.permute(2, 0, 1, 3).contiguous().view(Tq * B, D)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"]).view(Tq, B, D).transpose(1, 0)


def decoder_layer(sd: SD, i: int, x, mem, x_masks, cond_masks):
    p = f"layers.{i}."
    D = x.shape[-1]
    ln = lambda t, n: F.layer_norm(t, (D,), sd[p + n + ".weight"], sd[p + n + ".bias"], eps=1e-5)
    x = ln(x + mha(sd, p + "self_attn.", x, x, x_masks), "norm1")
    x = ln(x + mha(sd, p + "multihead_attn.", x, mem, cond_masks), "norm2")
    ff = F.linear(F.gelu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return ln(x + ff, "norm3")


def tfdec_forward(sd: SD, x, diffusion_step, conditioner, x_masks=None, cond_masks=None, *, num_layers=12):
    """convnext.py:331-379."""
    assert x.dim() == 3, f"mel must be 3 dim tensor, but got {x.dim()}"
    dim = sd["input_projection.2.weight"].shape[0]
    conv = lambda t, n: F.conv1d(t, sd[n + ".weight"], sd[n + ".bias"])
    x = conv(F.gelu(conv(x, "input_projection.0")), "input_projection.2").transpose(1, 2)
    x = x + sd["positional_embedding"][None, :x.size(1)] * sd["position_scale_query"]
    condition = conv(F.gelu(conv(conditioner, "condition_projection.0")), "condition_projection.2").transpose(1, 2)
    e = diffusion_embedding(diffusion_step, dim)
    e = F.linear(F.gelu(F.linear(e, sd["diffusion_embedding.1.weight"], sd["diffusion_embedding.1.bias"])),
                 sd["diffusion_embedding.3.weight"], sd["diffusion_embedding.3.bias"]).unsqueeze(1)
    condition = condition + sd["positional_embedding"][None, :condition.size(1)] * sd["position_scale_key"] + e
    if x_masks is not None:
        x = x.masked_fill(x_masks[..., None], 0.0)
    if cond_masks is not None:
        condition = condition.masked_fill(cond_masks[..., None], 0.0)
    for i in range(num_layers):
        x = decoder_layer(sd, i, x, condition, x_masks, cond_masks)
    x = x.transpose(1, 2)
    x = conv(F.gelu(conv(x, "output_projection.0")), "output_projection.2")
    if x_masks is not None:
        x = x.masked_fill(x_masks[:, None], 0.0)
    return x
