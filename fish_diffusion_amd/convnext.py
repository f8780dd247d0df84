"""`ConvNextDenoiser` on MI355X: same constructor, call signature and state-dict keys as
fish_diffusion/modules/convnext.py:155-262 (`ConvNext`, blocks :12-92), computed by libfishdx.so (csrc/convnext.hip).

Host plumbing only (parameters, repacking, C-ABI calls); there is no PyTorch fallback path.  `cross_attention=True`
(convnext.py:95-152,186-193: a `CrossAttentionBlock` -- an `nn.TransformerDecoderLayer` with 8 heads -- in front of every
`cross_every_n_layers`-th ConvNeXt block, which then run without the condition term) is built for dim 128 / 256 / 512.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib
from .registry import DENOISERS
from .wavenet import HipDenoiser, _attach


N_POSITIONS = 4096   # CrossAttentionBlock.get_embedding(num_embeddings=4096), convnext.py:114


def positional_embedding(dim: int, n: int = N_POSITIONS) -> torch.Tensor:
    """convnext.py:114-125 (the registered buffer of every CrossAttentionBlock)."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float) * -emb)
    emb = torch.arange(n, dtype=torch.float).unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1).view(n, -1)


def cross_block_table(p, dim, h):
    """A CrossAttentionBlock's tensors in its own state_dict order (own parameters, own buffer, then the children of
    nn.TransformerDecoderLayer, then diffusion_step_projection): (key, shape, init) with init = fan_in | "one" | "zero" | "pos" | "ln_w"."""
    rows = [(p + "position_scale_query", (1,), "one"), (p + "position_scale_key", (1,), "one"),
            (p + "positional_embedding", (N_POSITIONS, dim), "pos")]
    for att in ("self_attn", "multihead_attn"):
        rows += [(p + att + ".in_proj_weight", (3 * dim, dim), dim), (p + att + ".in_proj_bias", (3 * dim,), "zero"),
                 (p + att + ".out_proj.weight", (dim, dim), dim), (p + att + ".out_proj.bias", (dim,), "zero")]
    rows += [(p + "linear1.weight", (h, dim), dim), (p + "linear1.bias", (h,), dim),
             (p + "linear2.weight", (dim, h), h), (p + "linear2.bias", (dim,), h)]
    for n in ("norm1", "norm2", "norm3"):
        rows += [(p + n + ".weight", (dim,), "ln_w"), (p + n + ".bias", (dim,), "zero")]
    rows += [(p + "diffusion_step_projection.weight", (dim, dim, 1), dim), (p + "diffusion_step_projection.bias", (dim,), dim)]
    return rows


def param_table(mel_channels, dim, mlp_factor, condition_dim, num_layers, cross_every=0):
    """(state-dict key, shape, fan_in or None) in the module's registration order = the order fdx_convnext_pack expects.
    fan_in None marks the non-conv/linear parameters (gamma, LayerNorm affine).  cross_every > 0: `residual_layers` is the
    reference's mixed list [Cross, Block x cross_every, Cross, ...] (convnext.py:186-201) -- the indices count both kinds."""
    h = dim * mlp_factor
    rows = []

    def wb(prefix, out_c, in_c, conv):
        rows.append((prefix + ".weight", (out_c, in_c, 1) if conv else (out_c, in_c), in_c))
        rows.append((prefix + ".bias", (out_c,), in_c))

    wb("input_projection", dim, mel_channels, True)
    wb("diffusion_embedding.1", h, dim, False)
    wb("diffusion_embedding.3", dim, h, False)
    wb("conditioner_projection.0", h, condition_dim, True)
    wb("conditioner_projection.2", dim, h, True)
    j = 0
    for i in range(num_layers):
        if cross_every and i % cross_every == 0:
            rows += cross_block_table(f"residual_layers.{j}.", dim, h)
            j += 1
        p = f"residual_layers.{j}."
        j += 1
        rows.append((p + "gamma", (dim,), None))
        rows.append((p + "dwconv.weight", (dim, 1, 7), 7))
        rows.append((p + "dwconv.bias", (dim,), 7))
        rows.append((p + "norm.weight", (dim,), None))
        rows.append((p + "norm.bias", (dim,), None))
        wb(p + "pwconv1", h, dim, False)
        wb(p + "pwconv2", dim, h, False)
        wb(p + "diffusion_step_projection", dim, dim, True)
        wb(p + "condition_projection", dim, dim, True)
    wb("output_projection.0", dim, dim, True)
    wb("output_projection.2", mel_channels, dim, True)
    return rows


class ConvNext(HipDenoiser):
    """Drop-in for the reference `ConvNext` (registered as DENOISERS "ConvNextDenoiser")."""

    _KIND = "convnext"

    def __init__(self, mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=20, dilation_cycle=4,
                 gradient_checkpointing=False, cross_attention=False, cross_every_n_layers=5):
        super().__init__()
        if cross_attention and dim not in (128, 256, 512):
            raise NotImplementedError(f"ConvNext(cross_attention=True): the attention kernel is built for dim 128 / 256 / 512 (8 heads), got {dim}")
        self.mel_channels, self.dim, self.mlp_factor = mel_channels, dim, mlp_factor
        self.condition_dim, self.n_layers, self.dilation_cycle = condition_dim, num_layers, dilation_cycle
        self.gradient_checkpointing, self.cross_attention = gradient_checkpointing, cross_attention   # inference: unused
        self.cross_every_n_layers = cross_every_n_layers
        cross_every = int(cross_every_n_layers) if cross_attention else 0
        self._keys = []
        for key, shape, fan_in in param_table(mel_channels, dim, mlp_factor, condition_dim, num_layers, cross_every):
            t = torch.empty(shape)
            if fan_in == "pos":                    # a buffer in the reference (register_buffer, convnext.py:109): same state-dict key
                self._attach_buffer(key, positional_embedding(dim))
                self._keys.append(key)
                continue
            if fan_in == "one" or fan_in == "ln_w":
                t.fill_(1.0)
            elif fan_in == "zero":
                t.zero_()
            elif key.endswith("in_proj_weight"):
                nn.init.xavier_uniform_(t)         # nn.MultiheadAttention._reset_parameters
            elif key.endswith("gamma"):
                t.fill_(1e-6)                      # layer_scale_init_value, convnext.py:29,48
            elif key.endswith("norm.weight"):
                t.fill_(1.0)
            elif key.endswith("norm.bias"):
                t.zero_()
            else:                                  # nn.Conv1d / nn.Linear defaults: U(+-1/sqrt(fan_in)) for both
                bound = 1.0 / math.sqrt(fan_in)
                nn.init.uniform_(t, -bound, bound)
            _attach(self, key, nn.Parameter(t))
            self._keys.append(key)
        # the descriptor's last field is cross_every_n_layers (0 = no cross-attention)
        self._desc = _lib.ConvNextDesc(mel_channels, dim, mlp_factor, condition_dim, num_layers, int(dilation_cycle), cross_every)
        self._cond_channels = condition_dim
        self._init_engine()


    def _attach_buffer(self, dotted: str, value: torch.Tensor):
        *path, leaf = dotted.split(".")
        node = self
        for name in path:
            nxt = node._modules.get(name)
            if nxt is None:
                from .wavenet import _Group
                nxt = _Group()
                node.add_module(name, nxt)
            node = nxt
        node.register_buffer(leaf, value)


DENOISERS.register_module(name="ConvNextDenoiser", module=ConvNext, force=True)
DENOISERS.register_module(name="ConvNextDenoiserMI355X", module=ConvNext, force=True)
