"""`ConvNextDenoiser` on MI355X: same constructor, call signature and state-dict keys as
fish_diffusion/modules/convnext.py:155-262 (`ConvNext`, blocks :12-92), computed by libfishdx.so (csrc/convnext.hip).

Host plumbing only (parameters, repacking, C-ABI calls); there is no PyTorch fallback path.  The cross-attention variant
(`cross_attention=True`, convnext.py:95-152) is not built and raises at construction.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib
from .registry import DENOISERS
from .wavenet import HipDenoiser, _attach


def param_table(mel_channels, dim, mlp_factor, condition_dim, num_layers):
    """(state-dict key, shape, fan_in or None) in the module's registration order = the order fdx_convnext_pack expects.
    fan_in None marks the non-conv/linear parameters (gamma, LayerNorm affine)."""
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
    for i in range(num_layers):
        p = f"residual_layers.{i}."
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
        if cross_attention:
            raise NotImplementedError("ConvNext(cross_attention=True) is not built on the MI355X path")
        self.mel_channels, self.dim, self.mlp_factor = mel_channels, dim, mlp_factor
        self.condition_dim, self.n_layers, self.dilation_cycle = condition_dim, num_layers, dilation_cycle
        self.gradient_checkpointing, self.cross_attention = gradient_checkpointing, cross_attention   # inference: unused
        self._keys = []
        for key, shape, fan_in in param_table(mel_channels, dim, mlp_factor, condition_dim, num_layers):
            t = torch.empty(shape)
            if key.endswith("gamma"):
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
        self._desc = _lib.ConvNextDesc(mel_channels, dim, mlp_factor, condition_dim, num_layers, int(dilation_cycle), 0)
        self._cond_channels = condition_dim
        self._init_engine()


DENOISERS.register_module(name="ConvNextDenoiser", module=ConvNext, force=True)
DENOISERS.register_module(name="ConvNextDenoiserMI355X", module=ConvNext, force=True)
