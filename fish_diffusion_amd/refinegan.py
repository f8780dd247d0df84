"""RefineGAN vocoder on MI355X: `RefineGANGenerator`
(fish_diffusion/modules/vocoders/refinegan/generator.py:313-478) and the `RefineGAN` wrapper
(refinegan/refinegan.py:16-100, VOCODERS "RefineGAN") with the reference's constructor, methods and checkpoint keys; all
arithmetic is in libfishdx.so (`fdx_refinegan_forward`)."""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _lib
from .mel import PitchAdjustableMelSpectrogram
from .registry import VOCODERS
from .wavenet import _attach


def generator_param_table(cfg: dict):
    """(key, shape, weight_normed) in libfishdx's canonical order (include/fishdx.h); names from generator.py:333-423."""
    c = cfg["start_channels"]
    rows = []
    if cfg.get("template_generator", "comb") == "sine":   # SineGen.merge = Sequential(Linear(1, 1), Tanh), generator.py:233-236
        rows += [("template_gen.merge.0.weight", (1, 1), False), ("template_gen.merge.0.bias", (1,), False)]
    rows += [("template_conv.weight", (c, 1, 7), True), ("template_conv.bias", (c,), False)]
    for i, _ in enumerate(cfg["downsample_rates"]):
        n = 2 * c
        for j in range(3):
            p = f"downsample_blocks.{i}.1."
            rows += [(p + f"convs1.{j}.weight", (n, c if j == 0 else n, 7), True), (p + f"convs1.{j}.bias", (n,), False),
                     (p + f"convs2.{j}.weight", (n, n, 7), True), (p + f"convs2.{j}.bias", (n,), False)]
        c = n
    rows += [("mel_conv.weight", (c, cfg["num_mels"], 7), True), ("mel_conv.bias", (c,), False)]
    c *= 2
    sf0 = int(np.prod(cfg["upsample_rates"][1:]))
    rows += [("source_conv.weight", (c, 1, 2 * sf0), False), ("source_conv.bias", (c,), False)]
    for i, _ in enumerate(cfg["upsample_rates"]):
        n = c // 2
        p = f"upsample_conv_blocks.{i}."
        rows += [(p + "input_conv.weight", (n, c + c // 4, 7), False), (p + "input_conv.bias", (n,), False)]
        for b, k in enumerate((3, 7, 11)):
            rows += [(p + f"blocks.{b}.0.weight", (n,), False)]
            for j in range(3):
                rows += [(p + f"blocks.{b}.1.convs1.{j}.weight", (n, n, k), True), (p + f"blocks.{b}.1.convs1.{j}.bias", (n,), False),
                         (p + f"blocks.{b}.1.convs2.{j}.weight", (n, n, k), True), (p + f"blocks.{b}.1.convs2.{j}.bias", (n,), False)]
            rows += [(p + f"blocks.{b}.2.weight", (n,), False)]
        c = n
    rows += [("output_conv.weight", (1, c, 7), True), ("output_conv.bias", (1,), False)]
    return rows


class RefineGANGenerator(nn.Module):
    """Drop-in for generator.py `RefineGANGenerator(**config["generator"])`; parameters carry the reference's names, in
    weight-norm form until `remove_weight_norm()` folds them."""

    def __init__(self, *, sampling_rate: int = 44100, hop_length: int = 256, downsample_rates=(2, 2, 8, 8),
                 upsample_rates=(8, 8, 2, 2), leaky_relu_slope: float = 0.2, num_mels: int = 128, start_channels: int = 16,
                 template_generator: str = "comb"):
        super().__init__()
        if template_generator not in ("comb", "sine"):
            raise ValueError(f"Unknown template generator: {template_generator}")
        self.template_generator = template_generator
        assert np.prod(downsample_rates) == np.prod(upsample_rates) == hop_length
        self.sampling_rate, self.hop_length = sampling_rate, hop_length
        self.downsample_rates, self.upsample_rates = tuple(downsample_rates), tuple(upsample_rates)
        self.leaky_relu_slope = leaky_relu_slope
        self.cfg = dict(sampling_rate=sampling_rate, hop_length=hop_length, downsample_rates=self.downsample_rates,
                        upsample_rates=self.upsample_rates, leaky_relu_slope=leaky_relu_slope, num_mels=num_mels,
                        start_channels=start_channels, template_generator=template_generator)
        self._table = generator_param_table(self.cfg)
        self._weight_normed = True
        for key, shape, wn in self._table:
            if wn:
                v = torch.randn(shape) * 0.01
                g = v.flatten(1).norm(dim=1).view(-1, *([1] * (len(shape) - 1)))
                _attach(self, key + "_g", nn.Parameter(g))
                _attach(self, key + "_v", nn.Parameter(v))
            elif len(shape) == 1 and key.endswith("weight"):
                _attach(self, key, nn.Parameter(torch.ones(shape)))            # AdaIN weight (generator.py:96)
            else:
                _attach(self, key, nn.Parameter(torch.randn(shape) * 0.01 if key.endswith("weight") else torch.zeros(shape)))
        d = _lib.RefineGanDesc()
        d.sampling_rate, d.hop_length, d.num_mels, d.start_channels = sampling_rate, hop_length, num_mels, start_channels
        d.leaky_relu_slope = leaky_relu_slope
        d.template_sine = 1 if template_generator == "sine" else 0
        if len(self.downsample_rates) > _lib.MAX_STAGES:
            raise ValueError("too many stages")
        d.n_down, d.n_up = len(self.downsample_rates), len(self.upsample_rates)
        for i, r in enumerate(self.downsample_rates):
            d.downsample_rates[i] = int(r)
        for i, r in enumerate(self.upsample_rates):
            d.upsample_rates[i] = int(r)
        self._desc = d
        nb = C.c_size_t()
        _lib.check(_lib.lib().fdx_refinegan_packed_bytes(C.byref(d), C.byref(nb)))
        self.n_noises = _lib.lib().fdx_refinegan_num_noises(C.byref(d))
        self._handle: Optional[_lib.Handle] = None
        self._arena = None
        self._sig = None
        self.rng = "torch"   # "torch": noises drawn with torch.randn in the reference's order; "philox": on the device

    # ------------------------------------------------------------------ weights
    def remove_weight_norm(self) -> None:
        if not self._weight_normed:
            return
        sd = dict(self.named_parameters())
        for key, shape, wn in self._table:
            if not wn:
                continue
            g, v = sd[key + "_g"].data, sd[key + "_v"].data
            norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
            *path, leaf = key.split(".")
            node = self
            for name in path:
                node = node._modules[name]
            del node._parameters[leaf + "_g"], node._parameters[leaf + "_v"]
            node.register_parameter(leaf, nn.Parameter(v * (g / norm)))
        self._weight_normed = False

    def folded_weights(self) -> List[torch.Tensor]:
        sd = dict(self.named_parameters())
        out = []
        for key, shape, wn in self._table:
            if wn and self._weight_normed:
                g, v = sd[key + "_g"], sd[key + "_v"]
                norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
                out.append(v * (g / norm))
            else:
                out.append(sd[key])
        return out

    def load_folded_state(self, state: dict) -> None:
        self.remove_weight_norm()
        sd = dict(self.named_parameters())
        for key, shape, _ in self._table:
            if key not in state:
                raise KeyError(f"generator state is missing {key}")
            if tuple(state[key].shape) != tuple(shape):
                raise ValueError(f"{key}: expected {tuple(shape)}, got {tuple(state[key].shape)}")
            sd[key].data = state[key].detach().to(sd[key].device, torch.float32).clone()

    def engine(self, device: torch.device) -> _lib.Handle:
        device = torch.device("cuda", torch.cuda.current_device() if device.index is None else device.index)
        if self._handle is None or self._handle.device != device:
            self._handle = _lib.Handle(device)
            self._sig = None
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if sig != self._sig:
            with torch.no_grad():
                arena = _lib.pack_to_device(self._desc, self.folded_weights(), "refinegan", device)
            _lib.check(_lib.lib().fdx_refinegan_attach(self._handle.h, C.byref(self._desc), _lib.ptr(arena), arena.numel()), self._handle.h)
            self._arena, self._sig = arena, sig
        return self._handle

    def noise_shapes(self, B: int, T: int):
        """Shapes of the standard-normal draws one forward consumes, in the reference's order (generator.py:191,104-107)."""
        L = T * self.hop_length
        shapes = [(B, 1, L)]
        c = self.cfg["start_channels"] * 2 ** len(self.downsample_rates) * 2
        length = T
        for rate in self.upsample_rates:
            c //= 2
            length *= rate
            shapes += [(B, c, length)] * 6
        return shapes

    @torch.no_grad()
    def forward(self, mel: torch.Tensor, f0: torch.Tensor, noises: Optional[Sequence[torch.Tensor]] = None, mel_scale: float = 1.0):
        """mel [B, num_mels, T], f0 [B, 1, T] (or [B, T]) -> [B, 1, T * hop_length]  (generator.py:437-478)."""
        _lib.require_gpu(mel, "RefineGANGenerator input")
        if f0.dim() == 3:
            f0 = f0[:, 0]
        B, M, T = mel.shape
        if M != self.cfg["num_mels"] or tuple(f0.shape) != (B, T):
            raise ValueError(f"mel {tuple(mel.shape)} / f0 {tuple(f0.shape)} mismatch")
        eng = self.engine(mel.device)
        shapes = self.noise_shapes(B, T)
        seed = 0
        if noises is None and self.rng == "torch":
            if self.template_generator == "sine":   # SineGen draws (and then zeroes) the initial phases first: keep the RNG stream aligned
                torch.rand(B, 1, device=mel.device)   # generator.py:254-257
            noises = [torch.randn(s, device=mel.device) for s in shapes]   # same draw order as the reference's randn_like calls
        arr = None
        if noises is not None:
            if len(noises) != len(shapes) or any(tuple(n.shape) != s for n, s in zip(noises, shapes)):
                raise ValueError(f"expected {len(shapes)} noise tensors of shapes {shapes}")
            keep = [n.to(device=mel.device, dtype=torch.float32).contiguous() for n in noises]
            arr = (C.c_void_p * len(keep))(*[k.data_ptr() for k in keep])
        else:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        m = mel.to(torch.float32).contiguous()
        f = f0.to(device=mel.device, dtype=torch.float32).contiguous()
        wav = torch.empty((B, 1, T * self.hop_length), device=mel.device, dtype=torch.float32)
        with eng.lock:
            _lib.check(_lib.lib().fdx_refinegan_forward(eng.h, _lib.ptr(m), _lib.ptr(f), B, T, float(mel_scale), arr, seed,
                                                        _lib.ptr(wav), _lib.stream_ptr(mel.device)), eng.h)
        return wav


class RefineGAN(nn.Module):
    """Drop-in for refinegan.py `RefineGAN` (VOCODERS "RefineGAN"): checkpoint + config.json, `spec2wav`, `wav2spec`."""

    def __init__(self, checkpoint_path: str = "checkpoints/refinegan/model", config_file: Optional[str] = None,
                 use_natural_log: bool = True):
        super().__init__()
        if config_file is None:
            config_file = Path(checkpoint_path).parent / "config.json"
        with open(config_file) as f:
            config = json.loads(f.read())
        self.model = RefineGANGenerator(**config["generator"])
        self.use_natural_log, self.config = use_natural_log, config
        cp_dict = torch.load(checkpoint_path, map_location="cpu")
        if "state_dict" not in cp_dict:
            state = cp_dict["generator"]
        else:
            state = {k.replace("generator.", ""): v for k, v in cp_dict["state_dict"].items() if k.startswith("generator.")}
        self.model.load_state_dict(state)   # strict, weight-norm form (refinegan.py:38-49)
        self.model.eval()
        self.model.remove_weight_norm()
        self._finish()

    def _finish(self):
        c = self.config
        self.mel_transform = PitchAdjustableMelSpectrogram(sample_rate=c["sampling_rate"], n_fft=c["n_fft"], win_length=c["win_length"],
                                                           hop_length=c["hop_length"], f_min=c["f_min"], f_max=c["f_max"],
                                                           n_mels=c["num_mels"])

    @classmethod
    def from_state(cls, config: dict, folded_state: dict, use_natural_log: bool = True) -> "RefineGAN":
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.model = RefineGANGenerator(**config["generator"])
        self.use_natural_log, self.config = use_natural_log, config
        self.model.load_folded_state(folded_state)
        self.model.eval()
        self._finish()
        return self

    def freeze(self):
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()

    @property
    def device(self):
        return next(self.model.parameters()).device

    @torch.no_grad()
    def spec2wav(self, mel, f0, key_shift=0):
        """mel [num_mels, T], f0 [T] -> wav [T*hop]  (refinegan.py:67-78; f0 scaled in place like the reference)."""
        c = mel[None]
        f0 *= 2 ** (key_shift / 12)
        scale = 2.30259 if self.use_natural_log is False else 1.0
        return self.model(c, f0[None].to(c.dtype), mel_scale=scale).view(-1)

    @torch.no_grad()
    def wav2spec(self, wav_torch, sr=None, key_shift=0, speed=1.0):
        """refinegan.py:84-100 -- note the reference ignores key_shift / speed here (commented out at :96); so do we."""
        if sr is None:
            sr = self.config["sampling_rate"]
        if sr != self.config["sampling_rate"]:
            raise RuntimeError("resampling is host pre-processing (librosa) and is not part of the device path")
        mode = _lib.MEL_LN if self.use_natural_log is not False else _lib.MEL_LOG10
        return self.mel_transform(wav_torch, log_mode=mode)[0]


VOCODERS.register_module(name="RefineGAN", module=RefineGAN, force=True)
VOCODERS.register_module(name="RefineGANMI355X", module=RefineGAN, force=True)
