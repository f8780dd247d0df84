"""`WaveNetDenoiser` on MI355X: same constructor, call signature, state-dict keys and error behaviour as
fish_diffusion/modules/wavenet.py:151-236, computed by libfishdx.so (hand-written HIP, fp32 MFMA).

This module is host plumbing only: it owns the parameters (so `load_state_dict` of a reference
checkpoint works unchanged), repacks them into the kernels' fragment order when they change, and
forwards `data_ptr()`s to the C ABI.  There is no PyTorch fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Optional

import torch
from torch import nn

from . import _lib
from .registry import DENOISERS


class _Group(nn.Module):
    """Anonymous container: gives parameters the reference's dotted state-dict names."""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    *path, leaf = dotted.split(".")
    node = root
    for name in path:
        nxt = node._modules.get(name)
        if nxt is None:
            nxt = _Group()
            node.add_module(name, nxt)
        node = nxt
    node.register_parameter(leaf, param)


def param_table(mel_channels, d_encoder, residual_channels, residual_layers, use_linear_bias):
    """(state-dict key, shape, init) in the canonical order libfishdx expects (include/fishdx.h).
    init: 'kaiming' (ConvNorm, wavenet.py:75), 'xavier' (LinearNorm, :37), 'conv_bias' (nn.Conv1d default),
    'zero' (LinearNorm bias :39; final projection weight :192)."""
    C_ = residual_channels
    rows = [("input_projection.conv.weight", (C_, mel_channels, 1), "kaiming"),
            ("input_projection.conv.bias", (C_,), "conv_bias")]

    def linear(prefix, out_f, in_f):
        rows.append((prefix + ".linear.weight", (out_f, in_f), "xavier"))
        if use_linear_bias:
            rows.append((prefix + ".linear.bias", (out_f,), "zero"))

    def conv(prefix, out_c, in_c, k, init="kaiming"):
        rows.append((prefix + ".conv.weight", (out_c, in_c, k), init))
        rows.append((prefix + ".conv.bias", (out_c,), "conv_bias"))

    linear("mlp.0", 4 * C_, C_)
    linear("mlp.2", C_, 4 * C_)
    for i in range(residual_layers):
        p = f"residual_layers.{i}."
        conv(p + "conv_layer", 2 * C_, C_, 3)
        linear(p + "diffusion_projection", C_, C_)
        conv(p + "conditioner_projection", 2 * C_, d_encoder, 1)
        conv(p + "output_projection", 2 * C_, C_, 1)
    conv("skip_projection", C_, C_, 1)
    conv("output_projection", mel_channels, C_, 1, init="zero")
    return rows


class HipDenoiser(nn.Module):
    """Engine plumbing shared by the HIP denoisers (WaveNet here, ConvNext in convnext.py).  A subclass registers its
    parameters under the reference's state-dict names and sets `_keys` (canonical pack order), `_desc` (the C-ABI
    descriptor), `_KIND` (the `fdx_<kind>_*` entry-point family), `mel_channels` and `_cond_channels`."""

    _KIND = ""

    def _init_engine(self):
        # fail at construction (like a bad config would) rather than at first forward
        nb = C.c_size_t()
        _lib.check(self._fn("packed_bytes")(C.byref(self._desc), C.byref(nb)))
        self._handle: Optional[_lib.Handle] = None
        self._arena: Optional[torch.Tensor] = None
        self._packed_sig = None
        self._prep_sig = None
        self._prep_keep = None

    def _fn(self, name):
        return getattr(_lib.lib(), f"fdx_{self._KIND}_{name}")

    # ------------------------------------------------------------------ weights
    def _params(self):
        sd = dict(self.named_parameters())
        sd.update(self.named_buffers())          # e.g. the transformer denoiser's positional_embedding
        return [sd[k] for k in self._keys]

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self._params())

    def engine(self, device: torch.device) -> _lib.Handle:
        """The fdx handle for `device` with the current weights attached (repacked if they changed)."""
        device = torch.device("cuda", torch.cuda.current_device() if device.index is None else device.index)
        if self._handle is None or self._handle.device != device:
            self._handle = _lib.Handle(device)
            self._packed_sig = None
        sig = self._signature()
        if sig != self._packed_sig:
            self.attach_arena(_lib.pack_to_device(self._desc, self._params(), self._KIND, self._handle.device))
            self._packed_sig = sig
        return self._handle

    def attach_arena(self, arena: torch.Tensor):
        """Attach an already packed arena (e.g. one received by RCCL broadcast, see dist.py)."""
        if self._handle is None:
            self._handle = _lib.Handle(arena.device)
        _lib.check(self._fn("attach")(self._handle.h, C.byref(self._desc), _lib.ptr(arena), arena.numel()),
                   self._handle.h)
        self._arena = arena
        self._prep_sig = None
        self._packed_sig = self._signature()
        self._after_attach()

    def _after_attach(self):
        """Hook: extra arenas that ride on the main one (WaveNet's bf16 storage mode)."""

    def packed_arena(self, device) -> torch.Tensor:
        self.engine(torch.device(device))
        return self._arena

    # ------------------------------------------------------------------ step-invariant conditioner work
    def prepare(self, conditioner: torch.Tensor, cond_masks: Optional[torch.Tensor] = None) -> _lib.Handle:
        """Hoisted conditioner projections of all layers (wavenet.py:108 is step-invariant)."""
        _lib.require_gpu(conditioner, "denoiser conditioner")
        eng = self.engine(conditioner.device)
        # The hoisted work is cached on the identity of its inputs (address, version counter, geometry).  The module keeps the two
        # tensors alive while the cache entry is: otherwise the allocator could hand their addresses to a NEW conditioner
        # (version 0 again) and the stale slab would be reused silently.
        def ident(t):
            return None if t is None else (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype)
        with eng.lock:   # callers that go on to use the prepared state hold the same (re-entrant) lock across both steps
            # (the stream is part of the identity: work hoisted on one stream is not ordered before a run on another)
            sig = (ident(conditioner), ident(cond_masks), self._packed_sig, torch.cuda.current_stream(conditioner.device).cuda_stream)
            if sig != self._prep_sig:
                B, E, T = conditioner.shape
                if E != self._cond_channels:
                    raise ValueError(f"conditioner has {E} channels, expected {self._cond_channels}")
                cond = conditioner.to(torch.float32).contiguous()
                cm = None if cond_masks is None else cond_masks.to(torch.uint8).contiguous()
                _lib.check(self._fn("prepare")(eng.h, _lib.ptr(cond), B, T, _lib.ptr(cm), _lib.stream_ptr(cond.device)), eng.h)
                self._prep_sig = sig
                self._prep_keep = (conditioner, cond_masks)
        return eng

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, diffusion_step, conditioner, x_masks=None, cond_masks=None):
        """x [B, M, T] (or [B, 1, M, T]); diffusion_step [B] or [1] (long or float); conditioner [B, E, T]."""
        use_4_dim = False
        if x.dim() == 4:  # DiffSVC compatibility, wavenet.py:203-207
            x = x[:, 0]
            use_4_dim = True
        assert x.dim() == 3, f"mel must be 3 dim tensor, but got {x.dim()}"
        _lib.require_gpu(x, "denoiser input")
        B, M, T = x.shape
        if M != self.mel_channels or conditioner.shape[0] != B or conditioner.shape[2] != T:
            raise ValueError(f"x {tuple(x.shape)} does not match conditioner {tuple(conditioner.shape)}")
        xin = x.to(torch.float32).contiguous()
        t = diffusion_step.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        xm = None if x_masks is None else x_masks.to(torch.uint8).contiguous()
        out = torch.empty_like(xin)
        eng = self.engine(x.device)
        with eng.lock:   # prepare + forward as one critical section: another thread must not re-prepare the handle in between
            self.prepare(conditioner, cond_masks)
            _lib.check(self._fn("forward")(eng.h, _lib.ptr(xin), _lib.ptr(t), t.numel(), _lib.ptr(xm),
                                                      _lib.ptr(out), _lib.stream_ptr(x.device)), eng.h)
        return out[:, None] if use_4_dim else out


STORAGE_MODES = ("fp32", "bf16", "fp16x3")


class WaveNet(HipDenoiser):
    """Drop-in for the reference `WaveNet` (registered as DENOISERS "WaveNetDenoiser")."""

    _KIND = "wavenet"

    def __init__(self, mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=20,
                 use_linear_bias=False, dilation_cycle=None):
        super().__init__()
        self.mel_channels, self.d_encoder = mel_channels, d_encoder
        self.residual_channels, self.n_layers = residual_channels, residual_layers
        self.use_linear_bias, self.dilation_cycle = bool(use_linear_bias), dilation_cycle
        self._keys = []
        for key, shape, init in param_table(mel_channels, d_encoder, residual_channels, residual_layers,
                                            self.use_linear_bias):
            t = torch.empty(shape)
            if init == "kaiming":
                nn.init.kaiming_normal_(t)
            elif init == "xavier":
                nn.init.xavier_uniform_(t)
            elif init == "conv_bias":
                fan_in = {"input_projection.conv.bias": mel_channels}.get(key)
                if fan_in is None:
                    fan_in = d_encoder if "conditioner_projection" in key else residual_channels
                    if "conv_layer" in key:
                        fan_in *= 3
                bound = 1.0 / math.sqrt(fan_in)
                nn.init.uniform_(t, -bound, bound)
            else:
                nn.init.zeros_(t)
            _attach(self, key, nn.Parameter(t))
            self._keys.append(key)
        self._desc = _lib.WavenetDesc(mel_channels, d_encoder, residual_channels, residual_layers,
                                      int(dilation_cycle or 0), int(self.use_linear_bias))
        self._cond_channels = d_encoder
        self._storage = os.environ.get("FDX_WAVENET_STORAGE", "fp32")   # (the env default exists for the test suite: it re-runs the fp32 parity tests in the split mode)
        if self._storage not in STORAGE_MODES:
            raise ValueError(f"FDX_WAVENET_STORAGE must be one of {STORAGE_MODES}, got {self._storage!r}")
        self._arena_bf16 = None
        self._f16s_on = False
        self._init_engine()

    # ------------------------------------------------------------------ opt-in storage modes of the two residual-block GEMMs
    @property
    def storage(self) -> str:
        """"fp32" (default).
        "bf16": bf16 weights / activation operands, fp32 accumulate (BASELINE configs[4]).  Not parity-grade; DESIGN.md has its error.
        "fp16x3": every operand as an fp16 pair hi + lo, each product block hi.hi + hi.lo + lo.hi on the fp16 MFMA, fp32 accumulate:
        fp32-class results (the tests hold it to the fp32 path's own bars): 128-wide LDS tiles from 200 tiles per launch (batch >= 4 at
        10 s), 64 x 64 tiles below.  Nets with a dilation above 8 (dilation_cycle = 5) are refused: the kernels stage tile +/- 8 columns."""
        return self._storage

    @storage.setter
    def storage(self, mode: str):
        if mode not in STORAGE_MODES:
            raise ValueError(f"storage must be one of {STORAGE_MODES}, got {mode!r}")
        if mode != self._storage:
            prev, self._storage = self._storage, mode
            self._prep_sig = None
            if self._handle is not None and self._arena is not None:
                try:
                    self._after_attach()
                except Exception:       # e.g. fp16x3 on a net with a dilation beyond the kernels' LDS window: stay in the previous mode
                    self._storage = prev
                    self._after_attach()
                    raise

    def _after_attach(self):
        l = _lib.lib()
        if self._storage != "fp16x3" and self._f16s_on:
            _lib.check(l.fdx_wavenet_f16s_enable(self._handle.h, 0), self._handle.h)
            self._f16s_on = False
        if self._storage == "bf16":
            # Derived on the device from the ATTACHED fp32 arena, never from this module's own parameters: after
            # dist.broadcast_model_weights a non-source rank's parameters are whatever they were initialised with.
            nb = C.c_size_t()
            _lib.check(l.fdx_wavenet_bf16_packed_bytes(C.byref(self._desc), C.byref(nb)))
            dev = self._handle.device
            self._arena_bf16 = torch.empty(nb.value, dtype=torch.uint8, device=dev)
            _lib.check(l.fdx_wavenet_bf16_from_arena(self._handle.h, _lib.ptr(self._arena_bf16), nb.value, _lib.stream_ptr(dev)), self._handle.h)
            _lib.check(l.fdx_wavenet_bf16_attach(self._handle.h, _lib.ptr(self._arena_bf16), self._arena_bf16.numel()), self._handle.h)
        elif self._arena_bf16 is not None:
            _lib.check(l.fdx_wavenet_bf16_attach(self._handle.h, None, 0), self._handle.h)
            self._arena_bf16 = None
        if self._storage == "fp16x3":       # (after the bf16 arena is gone: the two modes exclude each other); weights derived from the attached arena
            _lib.check(l.fdx_wavenet_f16s_enable(self._handle.h, 1), self._handle.h)
            self._f16s_on = True


DENOISERS.register_module(name="WaveNetDenoiser", module=WaveNet, force=True)
DENOISERS.register_module(name="WaveNetDenoiserMI355X", module=WaveNet, force=True)
