"""The condition front end on MI355X: `DiffSinger.forward_features` (+ the inference call chain) for the
NaiveProjection encoders the SVC configs use -- SURVEY 8(f) row 1.

Mirrors, for inference:
  fish_diffusion/archs/diffsinger/diffsinger.py:20-134   `DiffSinger.__init__`, `get_mask_from_lengths`, `forward_features`
  fish_diffusion/modules/encoders/naive_projection.py:6-60 `NaiveProjectionEncoder` (same constructor, same parameter names)
  fish_diffusion/utils/pitch.py:12-22                     `pitch_to_scale`

`forward_features` is ONE fused HIP launch (`fdx_features_forward`): text Linear + speaker embedding / mix + pitch
Linear(pitch_to_scale(f0)) + pitch-shift / energy projections, added in the reference's order.  Encoders this module
cannot fuse (FastSpeech2 / BERT text encoders) raise NotImplementedError: they are not
on this path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import nn

from . import _lib
from .registry import DIFFUSIONS, Registry

ENCODERS = Registry("encoders")

F0_MIN, F0_MAX = 50.0, 1100.0   # utils/pitch.py:6-7


def pitch_to_scale(f0, f0_min=F0_MIN, f0_max=F0_MAX):
    """utils/pitch.py:12-22 (host / torch version; configs reference it as `preprocessing=pitch_to_scale`).
    Inside `DiffSinger.forward_features` the same arithmetic runs in the fused kernel."""
    f0_scale = (f0 - f0_min) / (f0_max - f0_min)
    f0_scale = f0_scale.clamp(0, 1)
    return f0_scale.unsqueeze(-1) if f0.ndim == 2 else f0_scale


@torch.no_grad()
def repeat_expand(content: torch.Tensor, target_len: int, mode: str = "nearest") -> torch.Tensor:
    """`fish_diffusion.utils.tensor.repeat_expand` (utils/tensor.py:7-43) for device tensors: `[S]`, `[C, S]` or `[B, C, S]` ->
    the same rank with `target_len` frames, F.interpolate(mode="nearest") index semantics.  (Inside the inference chain the
    expansion is fused into the front-end launch instead -- `DiffSinger.forward_features(..., mel_max_len=T)`.)"""
    if mode != "nearest":
        raise NotImplementedError(f"repeat_expand mode {mode!r}: only 'nearest' (the reference's default and only use) is built")
    if not torch.is_tensor(content):
        raise TypeError("repeat_expand on the MI355X path takes device tensors (the numpy branch of the reference is host preprocessing)")
    assert content.ndim in (1, 2, 3)
    _lib.require_gpu(content, "repeat_expand input")
    src = content.to(torch.float32).contiguous()
    S = src.shape[-1]
    rows = src.numel() // S
    out = torch.empty(src.shape[:-1] + (int(target_len),), device=src.device, dtype=torch.float32)
    h = _lib.Handle.shared(src.device)
    with h.lock:
        _lib.check(_lib.lib().fdx_repeat_expand(h.h, _lib.ptr(src), rows, S, int(target_len), _lib.ptr(out), _lib.stream_ptr(src.device)), h.h)
    return out.to(content.dtype) if content.dtype != torch.float32 and torch.is_floating_point(content) else out


def _is_pitch_to_scale(fn) -> bool:
    return fn is not None and getattr(fn, "__name__", "") == "pitch_to_scale"


class NaiveProjectionEncoder(nn.Module):
    """Parameter container with the reference's names (`projection.weight/bias`, `embedding.weight`); the arithmetic
    happens in `DiffSinger.forward_features`' fused launch, or -- when called on its own -- in the same kernel with
    this encoder as the only term."""

    def __init__(self, input_size, output_size, use_embedding: bool = False, use_neck: bool = False, neck_size: int = 8,
                 preprocessing=None):
        super().__init__()
        self.use_embedding, self.input_size, self.output_size = use_embedding, input_size, output_size
        self.use_neck, self.neck_size = bool(use_neck) and not use_embedding, int(neck_size)
        self.preprocessing = preprocessing
        if use_embedding:
            self.embedding = nn.Embedding(input_size, output_size)
            nn.init.normal_(self.embedding.weight, mean=0, std=output_size ** -0.5)
        elif use_neck:                                  # naive_projection.py:37-41: keys projection.0.* / projection.1.*
            if not 0 < self.neck_size <= _lib.MAX_NECK:
                raise ValueError(f"neck_size {neck_size}: the fused front end takes 1..{_lib.MAX_NECK}")
            self.projection = nn.Sequential(nn.Linear(input_size, self.neck_size), nn.Linear(self.neck_size, output_size))
        else:
            self.projection = nn.Linear(input_size, output_size)
        for m in self.modules():                        # reset_params, :48-55
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0.0)
        self._handle: Optional[_lib.Handle] = None

    def _engine(self, device):
        if self._handle is None or self._handle.device != device:
            self._handle = _lib.Handle(device)
        return self._handle

    def linear_params(self):
        """(w, b, neck, neck_w, neck_b) as fp32 contiguous tensors: the output Linear, and -- use_neck -- the bottleneck Linear."""
        f = lambda t: t.detach().to(torch.float32).contiguous()   # noqa: E731
        if self.use_neck:
            return f(self.projection[1].weight), f(self.projection[1].bias), self.neck_size, f(self.projection[0].weight), f(self.projection[0].bias)
        return f(self.projection.weight), f(self.projection.bias), 0, None, None

    @torch.no_grad()
    def forward(self, x, *args, **kwargs):
        """Stand-alone use (the reference calls encoders individually in a few tools): Linear over the last dim."""
        if self.use_embedding:
            raise NotImplementedError("stand-alone embedding lookup: use DiffSinger.forward_features")
        if self.preprocessing is not None:
            x = self.preprocessing(x)
        _lib.require_gpu(x, "NaiveProjectionEncoder input")
        lead = x.shape[:-1]
        x2 = x.to(torch.float32).reshape(1, -1, self.input_size).contiguous()
        out = torch.empty((1, x2.shape[1], self.output_size), device=x.device, dtype=torch.float32)
        eng = self._engine(x.device)
        w, b, neck, nw, nb = self.linear_params()
        with eng.lock:
            _lib.check(_lib.lib().fdx_features_forward_svs(eng.h, _lib.ptr(x2), 1, x2.shape[1], 0, x2.shape[1], self.input_size,
                                                           self.output_size, _lib.ptr(w), _lib.ptr(b), neck, _lib.ptr(nw) if neck else None,
                                                           _lib.ptr(nb) if neck else None, None, None, None, 0, _lib.ACT_NONE, None, 0,
                                                           _lib.ptr(out), _lib.stream_ptr(x.device)), eng.h)
        return out.reshape(*lead, self.output_size)


ENCODERS.register_module(name="NaiveProjectionEncoder", module=NaiveProjectionEncoder, force=True)


def _cfg_get(cfg, key):
    return cfg.get(key) if isinstance(cfg, dict) else getattr(cfg, key, None)


class DiffSinger(nn.Module):
    """`DiffSinger(model_config)` with `.text_encoder`, `.diffusion`, optional `.speaker_encoder`, `.pitch_encoder`,
    `.pitch_shift_encoder`, `.energy_encoder` -- the attribute names (hence checkpoint keys) of diffsinger.py:20-40."""

    def __init__(self, model_config):
        super().__init__()
        self.text_encoder = ENCODERS.build(dict(_cfg_get(model_config, "text_encoder")))
        self.diffusion = DIFFUSIONS.build(dict(_cfg_get(model_config, "diffusion")))
        for name in ("speaker_encoder", "pitch_encoder", "pitch_shift_encoder", "energy_encoder"):
            cfg = _cfg_get(model_config, name)
            if cfg:
                setattr(self, name, ENCODERS.build(dict(cfg)))
        self._handle: Optional[_lib.Handle] = None

    @staticmethod
    def get_mask_from_lengths(lengths, max_len=None):
        """True = padding (diffsinger.py:42-55)."""
        if max_len is None:
            max_len = int(torch.max(lengths).item())
        ids = torch.arange(0, int(max_len), device=lengths.device).unsqueeze(0).expand(lengths.shape[0], -1)
        return ids >= lengths.unsqueeze(1).expand(-1, int(max_len))

    def _engine(self, device):
        device = torch.device("cuda", torch.cuda.current_device() if device.index is None else device.index)
        if self._handle is None or self._handle.device != device:
            self._handle = _lib.Handle(device)
        return self._handle

    @staticmethod
    def _scalar_term(enc: NaiveProjectionEncoder, values: torch.Tensor, B: int, T: int, keep: list,
                     allow_expand: bool = False) -> _lib.FeatureTerm:
        if enc.use_embedding or enc.input_size != 1:
            raise NotImplementedError("only Linear(1 -> hidden) scalar encoders are fused")
        pre = _lib.PRE_NONE
        if enc.preprocessing is not None:
            if not _is_pitch_to_scale(enc.preprocessing):
                raise NotImplementedError(f"preprocessing {enc.preprocessing!r} is not fused (only pitch_to_scale)")
            pre = _lib.PRE_PITCH_TO_SCALE
        v = values.to(torch.float32)
        if v.ndim == 3 and v.shape[-1] == 1:
            v = v[..., 0]
        src_frames = 0
        if v.ndim == 2 and v.shape == (B, 1) and T != 1:
            v, per_frame = v[:, 0], 0
        elif v.ndim == 2 and v.shape == (B, T):
            per_frame = 1
        elif v.ndim == 2 and v.shape[0] == B and allow_expand:   # a track at another frame rate: fused repeat_expand
            per_frame, src_frames = 1, int(v.shape[1])
        elif v.ndim == 1 and v.shape[0] == B:
            per_frame = 0
        else:
            raise ValueError(f"scalar feature of shape {tuple(values.shape)} does not match batch {B} x frames {T}")
        v = v.contiguous()
        w, b, neck, nw, nb = enc.linear_params()
        w = w.reshape(-1) if not neck else w
        nw = nw.reshape(-1) if neck else None
        keep += [v, w, b, nw, nb]
        return _lib.FeatureTerm(_lib.TERM_SCALAR_LINEAR, per_frame, pre, src_frames, v.data_ptr(), w.data_ptr(), b.data_ptr(), F0_MIN, F0_MAX,
                                neck, nw.data_ptr() if neck else None, nb.data_ptr() if neck else None)

    @torch.no_grad()
    def forward_features(self, speakers, contents, contents_lens, contents_max_len, mel_lens=None, mel_max_len=None,
                         pitches=None, pitch_shift=None, phones2mel=None, energy=None, *, contents_channel_first=False,
                         expand_to: Optional[int] = None):
        """Reference signature (diffsinger.py:57-68) plus two keyword-only extensions that fuse the step before it in
        SVCInference.forward -- `repeat_expand(text_features, mel_len).T` (tools/diffusion/inference.py:113-114): with
        `expand_to=T` the contents (and a per-frame pitch track of another length) are read at their own frame rate and
        nearest-expanded to T frames inside the launch; `contents_channel_first` takes the extractor's `[B, Din, S]` layout."""
        if not isinstance(self.text_encoder, NaiveProjectionEncoder) or self.text_encoder.use_embedding:
            raise NotImplementedError("only the NaiveProjectionEncoder (Linear) text encoder is fused")
        _lib.require_gpu(contents, "contents")
        mel_masks = self.get_mask_from_lengths(mel_lens, mel_max_len) if mel_lens is not None else None
        if contents_channel_first:
            B, Din, S = contents.shape
        else:
            B, S, Din = contents.shape
        T = int(expand_to) if expand_to is not None else S
        if T <= 0:
            raise ValueError("expand_to must be positive")
        expand = expand_to is not None
        p2m = gmask = None
        if phones2mel is not None:        # SVS duration gather (diffsinger.py:85-90): frame t <- text frame phones2mel[b][t], * (1 - mel_mask)
            if expand:
                raise ValueError("phones2mel and expand_to are two different frame maps")
            if mel_masks is None:
                raise TypeError("phones2mel needs mel_lens: the reference multiplies the gathered features by 1 - mel_masks")
            p2m = phones2mel.to(device=contents.device, dtype=torch.int64).contiguous()
            if p2m.ndim != 2 or p2m.shape[0] != B:
                raise ValueError(f"phones2mel of shape {tuple(phones2mel.shape)} for a batch of {B}")
            T = int(p2m.shape[1])
            if tuple(mel_masks.shape) != (B, T):
                raise ValueError(f"mel mask {tuple(mel_masks.shape)} does not match phones2mel {tuple(p2m.shape)}")
            if T and (int(p2m.min()) < 0 or int(p2m.max()) >= S):
                raise RuntimeError("index out of range in phones2mel")    # torch.gather raises RuntimeError as well
            gmask = mel_masks.to(device=contents.device, dtype=torch.uint8).contiguous()
        E = self.text_encoder.output_size
        keep, terms = [], []
        # speaker: float embedding [B,E] / [B,T,E] given directly, or ids through speaker_encoder (diffsinger.py:92-108)
        if speakers is not None and speakers.ndim in (2, 3) and torch.is_floating_point(speakers):
            v = speakers.to(torch.float32).contiguous()
            if v.shape[-1] != E or (v.ndim == 3 and v.shape[1] not in (1, T)):
                raise ValueError(f"speaker embedding {tuple(v.shape)} does not broadcast to [{B}, {T}, {E}]")
            per_frame = int(v.ndim == 3 and v.shape[1] == T and T != 1)
            keep.append(v)
            terms.append(_lib.FeatureTerm(_lib.TERM_VECTOR, per_frame, 0, 0, v.data_ptr(), None, None, 0.0, 0.0, 0, None, None))
        elif speakers is not None and hasattr(self, "speaker_encoder"):
            enc = self.speaker_encoder
            if not enc.use_embedding:
                raise NotImplementedError("only the embedding speaker encoder is fused")
            ids = speakers.to(torch.int64).reshape(-1).contiguous()
            if ids.numel() != B:
                raise ValueError(f"expected one speaker id per utterance, got {tuple(speakers.shape)}")
            if int(ids.min()) < 0 or int(ids.max()) >= enc.input_size:
                raise IndexError("speaker id out of range")   # nn.Embedding raises IndexError as well
            tab = enc.embedding.weight.detach().to(torch.float32).contiguous()
            keep += [ids, tab]
            terms.append(_lib.FeatureTerm(_lib.TERM_EMBEDDING, 0, 0, 0, ids.data_ptr(), tab.data_ptr(), None, 0.0, 0.0, 0, None, None))
        if hasattr(self, "pitch_encoder"):
            terms.append(self._scalar_term(self.pitch_encoder, pitches, B, T, keep, allow_expand=expand))
        if pitch_shift is not None and hasattr(self, "pitch_shift_encoder"):
            terms.append(self._scalar_term(self.pitch_shift_encoder, pitch_shift, B, T, keep))
        if energy is not None and hasattr(self, "energy_encoder"):
            terms.append(self._scalar_term(self.energy_encoder, energy, B, T, keep, allow_expand=expand))
        if len(terms) > _lib.MAX_FEATURE_TERMS:
            raise ValueError("too many additive terms")

        x = contents.to(torch.float32).contiguous()
        w, b, neck, nw, nb = self.text_encoder.linear_params()
        out = torch.empty((B, T, E), device=contents.device, dtype=torch.float32)
        arr = (_lib.FeatureTerm * max(1, len(terms)))(*terms)
        eng = self._engine(contents.device)
        with eng.lock:
            _lib.check(_lib.lib().fdx_features_forward_svs(eng.h, _lib.ptr(x), B, S, int(bool(contents_channel_first)), T, Din, E,
                                                           _lib.ptr(w), _lib.ptr(b), neck, _lib.ptr(nw) if neck else None,
                                                           _lib.ptr(nb) if neck else None, _lib.ptr(p2m) if p2m is not None else None,
                                                           _lib.ptr(gmask) if gmask is not None else None, arr, len(terms), _lib.ACT_NONE,
                                                           None, 0, _lib.ptr(out), _lib.stream_ptr(contents.device)), eng.h)
        del p2m, gmask, nw, nb
        del keep
        return dict(features=out, x_masks=mel_masks, x_lens=mel_lens, cond_masks=mel_masks)

    def forward(self, *a, **k):
        raise NotImplementedError("DiffSinger.forward is the training step (diffsinger.py:136-179); inference calls "
                                  "forward_features(...) then .diffusion(features, ...) (tools/diffusion/inference.py:140-159)")

    @torch.no_grad()
    def infer(self, speakers, contents, pitches, *, sampler_interval=None, noise_predictor=None, skip_steps=0,
              original_mel=None, pitch_shift=None, energy=None, mel_lens=None, mel_len: Optional[int] = None,
              contents_channel_first: bool = False, **diffusion_kwargs):
        """The call chain of SVCInference.forward (tools/diffusion/inference.py:104-159) for a batch: extractor output -> mel.
        `contents` [B, T, Din]; or, with `mel_len=T` (`audio.shape[-1] // 512`, :104), the extractor's own frames `[B, S, Din]` /
        `[B, Din, S]` (`contents_channel_first`) nearest-expanded to T inside the front-end launch (:113-114), as is a pitch
        track given at another length (:108-109)."""
        B = contents.shape[0]
        S = contents.shape[2 if contents_channel_first else 1]
        T = int(mel_len) if mel_len is not None else S
        lens = mel_lens if mel_lens is not None else torch.full((B,), T, device=contents.device, dtype=torch.long)
        f = self.forward_features(speakers=speakers, contents=contents, contents_lens=lens, contents_max_len=T, mel_lens=mel_lens,
                                  mel_max_len=T if mel_lens is not None else None, pitches=pitches, pitch_shift=pitch_shift,
                                  energy=energy, contents_channel_first=contents_channel_first,
                                  expand_to=T if (mel_len is not None or contents_channel_first) else None)
        return self.diffusion(f["features"], sampler_interval=sampler_interval, noise_predictor=noise_predictor,
                              skip_steps=skip_steps, original_mel=original_mel, x_masks=f["x_masks"], cond_masks=f["cond_masks"],
                              **diffusion_kwargs)
