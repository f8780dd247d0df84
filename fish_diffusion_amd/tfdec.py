"""`TransformerDecoderDenoiser` on MI355X: same constructor, call signature and state-dict keys as
fish_diffusion/modules/convnext.py:263-379 (12 x nn.TransformerDecoderLayer between 1x1-conv projections), computed by
libfishdx.so (csrc/tfdec.hip: convgemm GEMMs + an fp32 MFMA flash-attention kernel + LayerNorm).

Host plumbing only; there is no PyTorch fallback path.  The conditioner must have as many frames as the mel (how
`GaussianDiffusion` calls every denoiser, diffusion.py:217); at most 4096 frames (the positional table, convnext.py:317).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib
from .registry import DENOISERS
from .wavenet import HipDenoiser, _attach

N_POSITIONS = 4096


def positional_embedding(dim: int, n: int = N_POSITIONS) -> torch.Tensor:
    """convnext.py:317-329 (a registered buffer: checkpoints carry their own copy)."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float) * -emb)
    emb = torch.arange(n, dtype=torch.float).unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1).view(n, -1)


def param_table(mel_channels, dim, mlp_factor, condition_dim, num_layers):
    """(state-dict key, shape, init) in state_dict order = the order fdx_tfdec_pack expects.  init: fan-in for the default
    nn.Conv1d / nn.Linear bound, 'xavier' (MultiheadAttention.in_proj_weight), 'zero', 'one', 'pos' (the buffer)."""
    h = dim * mlp_factor
    rows = [("position_scale_query", (1,), "one"), ("position_scale_key", (1,), "one"), ("positional_embedding", (N_POSITIONS, dim), "pos")]

    def wb(prefix, out_c, in_c, conv, bias_init=None):
        rows.append((prefix + ".weight", (out_c, in_c, 1) if conv else (out_c, in_c), in_c))
        rows.append((prefix + ".bias", (out_c,), bias_init if bias_init is not None else in_c))

    wb("input_projection.0", h, mel_channels, True)
    wb("input_projection.2", dim, h, True)
    wb("diffusion_embedding.1", h, dim, False)
    wb("diffusion_embedding.3", dim, h, False)
    wb("condition_projection.0", h, condition_dim, True)
    wb("condition_projection.2", dim, h, True)
    for i in range(num_layers):
        p = f"layers.{i}."
        for att in ("self_attn", "multihead_attn"):
            rows.append((p + att + ".in_proj_weight", (3 * dim, dim), "xavier"))
            rows.append((p + att + ".in_proj_bias", (3 * dim,), "zero"))
            wb(p + att + ".out_proj", dim, dim, False, bias_init="zero")
        wb(p + "linear1", h, dim, False)
        wb(p + "linear2", dim, h, False)
        for nrm in ("norm1", "norm2", "norm3"):
            rows.append((p + nrm + ".weight", (dim,), "one"))
            rows.append((p + nrm + ".bias", (dim,), "zero"))
    wb("output_projection.0", dim, dim, True)
    wb("output_projection.2", mel_channels, dim, True)
    return rows


class TransformerDecoderDenoiser(HipDenoiser):
    """Drop-in for the reference class of the same name (DENOISERS "TransformerDecoderDenoiser")."""

    _KIND = "tfdec"

    def __init__(self, mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=12, gradient_checkpointing=False):
        super().__init__()
        self.mel_channels, self.dim, self.mlp_factor = mel_channels, dim, mlp_factor
        self.condition_dim, self.n_layers = condition_dim, num_layers
        self.gradient_checkpointing = gradient_checkpointing   # inference: unused
        self._keys = []
        for key, shape, init in param_table(mel_channels, dim, mlp_factor, condition_dim, num_layers):
            if init == "pos":
                self.register_buffer(key, positional_embedding(dim))
                self._keys.append(key)
                continue
            t = torch.empty(shape)
            if init == "one":
                t.fill_(1.0)
            elif init == "zero":
                t.zero_()
            elif init == "xavier":
                nn.init.xavier_uniform_(t)
            else:
                bound = 1.0 / math.sqrt(init)
                nn.init.uniform_(t, -bound, bound)
            _attach(self, key, nn.Parameter(t))
            self._keys.append(key)
        self._desc = _lib.TfdecDesc(mel_channels, dim, mlp_factor, condition_dim, num_layers, N_POSITIONS)
        self._cond_channels = condition_dim
        self._init_engine()

    @torch.no_grad()
    def forward(self, x, diffusion_step, conditioner, x_masks=None, cond_masks=None):
        assert x.dim() == 3, f"mel must be 3 dim tensor, but got {x.dim()}"   # convnext.py:341 (no 4-D DiffSVC form here)
        if x.shape[-1] > N_POSITIONS:
            raise ValueError(f"{x.shape[-1]} frames exceed the positional table ({N_POSITIONS})")
        return super().forward(x, diffusion_step, conditioner, x_masks, cond_masks)


DENOISERS.register_module(name="TransformerDecoderDenoiser", module=TransformerDecoderDenoiser, force=True)
DENOISERS.register_module(name="TransformerDecoderDenoiserMI355X", module=TransformerDecoderDenoiser, force=True)
