"""NSF-HiFiGAN vocoder on MI355X: `Generator` (modules/vocoders/nsf_hifigan/models.py:353-448) and the
`NsfHifiGAN` wrapper (nsf_hifigan.py:16-107, VOCODERS "NsfHifiGAN") with the reference's constructor,
methods and checkpoint keys; all arithmetic is in libfishdx.so (`fdx_nsf_forward`, `fdx_mel_forward`)."""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _lib
from .mel import PitchAdjustableMelSpectrogram
from .registry import VOCODERS
from .wavenet import _Group, _attach


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def generator_param_table(h):
    """(key, shape, weight_normed) in libfishdx's canonical order (include/fishdx.h); names from models.py:362-403."""
    C0 = h["upsample_initial_channel"]
    rates, ksz = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
    rows = [("m_source.l_linear.weight", (1, 9), False), ("m_source.l_linear.bias", (1,), False),
            ("conv_pre.weight", (C0, h["num_mels"], 7), True), ("conv_pre.bias", (C0,), False)]
    for i, (u, k) in enumerate(zip(rates, ksz)):
        cin, cout = C0 // (2 ** i), C0 // (2 ** (i + 1))
        rows += [(f"ups.{i}.weight", (cin, cout, k), True), (f"ups.{i}.bias", (cout,), False)]
        nk = 2 * int(np.prod(rates[i + 1:])) if i + 1 < len(rates) else 1
        rows += [(f"noise_convs.{i}.weight", (cout, 1, nk), False), (f"noise_convs.{i}.bias", (cout,), False)]
    n = 0
    for i in range(len(rates)):
        ch = C0 // (2 ** (i + 1))
        for k, dils in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
            groups = ("convs1", "convs2") if str(h["resblock"]) == "1" else ("convs",)
            for j in range(len(dils)):
                for grp in groups:
                    rows += [(f"resblocks.{n}.{grp}.{j}.weight", (ch, ch, k), True),
                             (f"resblocks.{n}.{grp}.{j}.bias", (ch,), False)]
            n += 1
    rows += [("conv_post.weight", (1, ch, 7), True), ("conv_post.bias", (1,), False)]
    return rows


def make_desc(h) -> _lib.NsfDesc:
    d = _lib.NsfDesc()
    rates, ksz = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
    rk, rd = list(h["resblock_kernel_sizes"]), [list(x) for x in h["resblock_dilation_sizes"]]
    if len(rates) > _lib.MAX_STAGES or len(rk) > _lib.MAX_RESK or any(len(x) > _lib.MAX_DIL for x in rd):
        raise ValueError("generator config exceeds libfishdx limits")
    if len({len(x) for x in rd}) != 1:
        raise ValueError("all resblock_dilation_sizes rows must have the same length")
    d.num_mels, d.upsample_initial_channel, d.n_stages = h["num_mels"], h["upsample_initial_channel"], len(rates)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        d.upsample_rates[i], d.upsample_kernel_sizes[i] = u, k
    d.n_resblock_kernels, d.n_dilations = len(rk), len(rd[0])
    for j, k in enumerate(rk):
        d.resblock_kernel_sizes[j] = k
        for q, dil in enumerate(rd[j]):
            d.resblock_dilations[j][q] = dil
    d.resblock_type = int(h["resblock"])
    d.sampling_rate, d.hop_size, d.harmonic_num = h["sampling_rate"], h["hop_size"], 8
    return d


class Generator(nn.Module):
    """Drop-in for models.py `Generator(h)`; parameters carry the reference's names, in weight-norm form
    (`weight_g` / `weight_v`) until `remove_weight_norm()` folds them, exactly like the reference."""

    def __init__(self, h):
        super().__init__()
        self.h = h if isinstance(h, AttrDict) else AttrDict(dict(h))
        self.num_kernels = len(self.h["resblock_kernel_sizes"])
        self.num_upsamples = len(self.h["upsample_rates"])
        self._table = generator_param_table(self.h)
        self._weight_normed = True
        for key, shape, wn in self._table:
            if wn:
                v = torch.randn(shape) * 0.01
                g = v.flatten(1).norm(dim=1).view(-1, *([1] * (len(shape) - 1)))
                _attach(self, key + "_g", nn.Parameter(g))
                _attach(self, key + "_v", nn.Parameter(v))
            else:
                _attach(self, key, nn.Parameter(torch.randn(shape) * 0.01 if key.endswith("weight") else torch.zeros(shape)))
        self._desc = make_desc(self.h)
        nb = C.c_size_t()
        _lib.check(_lib.lib().fdx_nsf_packed_bytes(C.byref(self._desc), C.byref(nb)))
        self._handle: Optional[_lib.Handle] = None
        self._arena = None
        self._sig = None
        self._lanes: list = []       # [(Handle, torch.cuda.Stream)]: extra engines over the SAME packed arena, one stream each
        self._lanes_arena = None
        self.rng = "torch"  # "torch": draw rand_ini / source noise with torch (reference RNG order); "philox": on device

    def remove_weight_norm(self):
        """Fold g * v / ||v|| into `.weight` (torch.nn.utils.remove_weight_norm semantics, models.py:440-448)."""
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

    def folded_weights(self):
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

    def load_folded_state(self, state):
        """Load a state dict whose weight-normed layers are already folded (`.weight` keys)."""
        self.remove_weight_norm()
        missing = [k for k, _, _ in self._table if k not in state]
        if missing:
            raise KeyError(f"generator state is missing {missing[:4]}...")
        sd = dict(self.named_parameters())
        for key, shape, _ in self._table:
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
                arena = _lib.pack_to_device(self._desc, self.folded_weights(), "nsf", device)
            self.attach_arena(arena)
            self._sig = sig
        return self._handle

    def attach_arena(self, arena: torch.Tensor):
        if self._handle is None:
            self._handle = _lib.Handle(arena.device)
        _lib.check(_lib.lib().fdx_nsf_attach(self._handle.h, C.byref(self._desc), _lib.ptr(arena), arena.numel()), self._handle.h)
        self._arena = arena
        self._sig = tuple((p.data_ptr(), p._version) for p in self.parameters())

    def lanes(self, device: torch.device, n: int):
        """`n` extra engines sharing this module's packed arena, each with a workspace and a HIP stream of its own: independent
        utterances go through the generator side by side (a batch-1 pass leaves part of the chip idle in the small-grid stages;
        every pass is still the batch-1 pass, so each waveform is bit-identical to a call alone).  A lane's engine is only ever
        used on the lane's stream -- the workspace of a handle is ordered by its stream.
        Memory: a lane's workspace grows to the longest utterance it has vocoded (~0.9 GB per 10 s at hop 512: every stage's activations
        live at once), so `n` lanes hold ~n times the vocoder workspace until `release_lanes()`; `pipeline.synthesize` sizes `n` from the
        micro-batch and from free device memory."""
        primary = self.engine(torch.device(device))
        if self._lanes_arena is not self._arena or (self._lanes and self._lanes[0][0].device != primary.device):
            self._lanes, self._lanes_arena = [], self._arena
        while len(self._lanes) < n:
            hnd = _lib.Handle(primary.device)
            _lib.check(_lib.lib().fdx_nsf_attach(hnd.h, C.byref(self._desc), _lib.ptr(self._arena), self._arena.numel()), hnd.h)
            self._lanes.append((hnd, torch.cuda.Stream(device=primary.device)))
        return self._lanes[:n]

    def release_lanes(self):
        """Drop the extra engines `lanes()` created (their workspaces and streams); the next `lanes()` call builds them again."""
        self._lanes, self._lanes_arena = [], None

    def packed_arena(self, device) -> torch.Tensor:
        self.engine(torch.device(device))
        return self._arena

    @torch.no_grad()
    def forward(self, x, f0, rand_ini=None, src_noise=None, mel_scale: float = 1.0, engine: Optional[_lib.Handle] = None):
        """x [B, num_mels, T] (natural-log mel), f0 [B, T] or [B, 1, T] -> wav [B, 1, T*hop] (models.py:407-438).
        `engine`: one of `lanes()`'s handles (the caller is inside `torch.cuda.stream(lane_stream)`); default: the module's own."""
        _lib.require_gpu(x, "Generator input")
        if f0.dim() == 3:
            f0 = f0[:, 0]
        B, M, T = x.shape
        if M != self.h["num_mels"] or tuple(f0.shape) != (B, T):
            raise ValueError(f"mel {tuple(x.shape)} / f0 {tuple(f0.shape)} mismatch")
        eng = self.engine(x.device) if engine is None else engine
        L = T * self.h["hop_size"]
        mel = x.to(torch.float32).contiguous()
        f0c = f0.to(device=x.device, dtype=torch.float32).contiguous()
        seed = 0
        if self.rng == "torch":
            if rand_ini is None:  # models.py:210 then :289 -- same draw order as the reference
                rand_ini = torch.rand(B, 9, device=x.device)
            if src_noise is None:
                src_noise = torch.randn(B, L, 9, device=x.device)
        elif rand_ini is None or src_noise is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        ri = None if rand_ini is None else rand_ini.to(torch.float32).contiguous()
        sn = None if src_noise is None else src_noise.to(torch.float32).contiguous()
        wav = torch.empty((B, 1, L), device=x.device, dtype=torch.float32)
        with eng.lock:
            _lib.check(_lib.lib().fdx_nsf_forward(eng.h, _lib.ptr(mel), _lib.ptr(f0c), B, T, float(mel_scale), _lib.ptr(ri),
                                                  _lib.ptr(sn), seed, _lib.ptr(wav), _lib.stream_ptr(x.device)), eng.h)
        return wav

    @torch.no_grad()
    def source(self, f0, rand_ini, src_noise):
        """Harmonic source only (models.py:411-416) -> [B, 1, T*hop].  Test hook."""
        _lib.require_gpu(f0, "f0")
        B, T = f0.shape
        eng = self.engine(f0.device)
        L = T * self.h["hop_size"]
        har = torch.empty((B, 1, L), device=f0.device, dtype=torch.float32)
        with eng.lock:
            _lib.check(_lib.lib().fdx_nsf_source(eng.h, _lib.ptr(f0.float().contiguous()), B, T, _lib.ptr(rand_ini.float().contiguous()),
                                                 _lib.ptr(src_noise.float().contiguous()), 0, _lib.ptr(har),
                                                 _lib.stream_ptr(f0.device)), eng.h)
        return har


class NsfHifiGAN(nn.Module):
    """Drop-in for nsf_hifigan.py `NsfHifiGAN` (a LightningModule there; `.freeze()`, `.device`, `.h`, `.model`,
    `spec2wav`, `wav2spec` are what callers use: tools/diffusion/inference.py:76-78,99,160)."""

    def __init__(self, checkpoint_path: str = "checkpoints/nsf_hifigan/model", config_file: Optional[str] = None,
                 use_natural_log: bool = True, **kwargs):
        super().__init__()
        if config_file is None:
            config_file = Path(checkpoint_path).parent / "config.json"
        with open(config_file) as f:
            self.h = AttrDict(json.loads(f.read()))
        self.model = Generator(self.h)
        self.use_natural_log = use_natural_log
        cp_dict = torch.load(checkpoint_path, map_location="cpu")
        if "state_dict" not in cp_dict:
            state = cp_dict["generator"]
        else:
            state = {k.replace("generator.", ""): v for k, v in cp_dict["state_dict"].items() if k.startswith("generator.")}
        self.model.load_state_dict(state)  # strict, weight-norm form (nsf_hifigan.py:38-49)
        self.model.eval()
        self.model.remove_weight_norm()
        self._finish(kwargs)

    def _finish(self, kwargs):
        self.mel_transform = PitchAdjustableMelSpectrogram(
            sample_rate=self.h.sampling_rate, n_fft=self.h.n_fft, win_length=self.h.win_size, hop_length=self.h.hop_size,
            f_min=self.h.fmin, f_max=self.h.fmax, n_mels=self.h.num_mels)
        if "mel_channels" in kwargs:
            kwargs["num_mels"] = kwargs.pop("mel_channels")
        for k, v in kwargs.items():  # nsf_hifigan.py:64-70
            if getattr(self.h, k, None) != v:
                raise ValueError(f"Incorrect value for {k}: {v}")

    @classmethod
    def from_state(cls, h: dict, folded_state: dict, use_natural_log: bool = True, **kwargs) -> "NsfHifiGAN":
        """Build from an in-memory config + folded generator state (no files): synthetic-weight tests and bench."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.h = AttrDict(dict(h))
        self.model = Generator(self.h)
        self.use_natural_log = use_natural_log
        self.model.load_folded_state(folded_state)
        self.model.eval()
        self._finish(dict(kwargs))
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
        """mel [num_mels, T], f0 [T] -> wav [T*hop]  (nsf_hifigan.py:72-85)."""
        c = mel[None]
        if key_shift is not None and key_shift != 0:
            f0 *= 2 ** (key_shift / 12)  # in place, like the reference
        scale = 2.30259 if self.use_natural_log is False else 1.0
        f0 = f0[None].to(c.dtype)
        return self.model(c, f0, mel_scale=scale).view(-1)

    @torch.no_grad()
    def wav2spec(self, wav_torch, sr=None, key_shift=0, speed=1.0):
        """wav [1, N] -> log-mel [num_mels, T]  (nsf_hifigan.py:91-107)."""
        if sr is None:
            sr = self.h.sampling_rate
        if sr != self.h.sampling_rate:
            try:
                import librosa  # host-side resampling, as in the reference (:95-99); not part of the device path
            except ImportError as e:
                raise RuntimeError(f"resampling {sr} -> {self.h.sampling_rate} Hz needs librosa (host pre-processing)") from e
            res = librosa.resample(wav_torch.cpu().numpy(), orig_sr=sr, target_sr=self.h.sampling_rate)
            wav_torch = torch.from_numpy(res).to(wav_torch.device)
        mode = _lib.MEL_LN if self.use_natural_log is not False else _lib.MEL_LOG10
        return self.mel_transform(wav_torch, key_shift=key_shift, speed=speed, log_mode=mode)[0]


VOCODERS.register_module(name="NsfHifiGAN", module=NsfHifiGAN, force=True)
VOCODERS.register_module(name="NsfHifiGANMI355X", module=NsfHifiGAN, force=True)
