"""HiFiSinger (what configs/svc_hifisinger_v2.py runs) on MI355X: encoders -> feature_fuser -> RefineGAN generator,
mirroring fish_diffusion/archs/hifisinger/core.py:9-141 for inference.

forward_features = three fused launches of the front-end kernel (`fdx_features_forward_ex`):
    text Linear + speaker embedding (+ pitch-shift / energy projections)      core.py:70-105
    feature_fuser[0] Linear + SiLU                                            core.py:24-29,107
    feature_fuser[2] Linear + SiLU, `*= 1 - src_masks`, written channel-first core.py:107-110 (and the transpose of :137)
forward = forward_features + `RefineGANGenerator(features, pitches)` (core.py:136-139).
Both encoder variants are built: RefineGAN (hifi_svc_v2) and the NSF-HiFiGAN generator with num_mels = hidden_size
(hifi_svc v1, core.py:35-37,140-141).  The phones2mel gather of SVS (core.py:72-78) is part of the first launch.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import _lib
from .diffsinger import ENCODERS, DiffSinger, NaiveProjectionEncoder, _cfg_get
from .nsf_hifigan import AttrDict, Generator
from .refinegan import RefineGANGenerator


class HiFiSinger(nn.Module):
    def __init__(self, model_config):
        super().__init__()
        self.text_encoder = ENCODERS.build(dict(_cfg_get(model_config, "text_encoder")))
        self.speaker_encoder = ENCODERS.build(dict(_cfg_get(model_config, "speaker_encoder")))
        for name in ("pitch_shift_encoder", "energy_encoder"):
            cfg = _cfg_get(model_config, name)
            if cfg:
                setattr(self, name, ENCODERS.build(dict(cfg)))
        hidden = _cfg_get(model_config, "hidden_size")
        self.feature_fuser = nn.Sequential(nn.Linear(hidden, hidden), nn.SiLU(), nn.Linear(hidden, hidden), nn.SiLU())
        enc = dict(_cfg_get(model_config, "encoder"))
        if enc.get("type") == "RefineGAN":        # hifi_svc_v2 (core.py:31-34)
            enc.pop("type")
            self.encoder_type = "RefineGAN"
            self.encoder = RefineGANGenerator(**enc)
        else:                                      # hifi_svc v1: the NSF-HiFiGAN generator fed with the fused features (core.py:35-37)
            self.encoder_type = "HiFiGAN"
            self.encoder = Generator(AttrDict(enc))
        self._handle: Optional[_lib.Handle] = None

    get_mask_from_lengths = staticmethod(DiffSinger.get_mask_from_lengths)

    def _engine(self, device):
        device = torch.device("cuda", torch.cuda.current_device() if device.index is None else device.index)
        if self._handle is None or self._handle.device != device:
            self._handle = _lib.Handle(device)
        return self._handle

    def _launch(self, eng, x, lin, terms, act, mask, channel_first, gather=None, gather_mask=None):
        """One front-end launch.  `lin` = (w, b, neck, neck_w, neck_b) as NaiveProjectionEncoder.linear_params() returns them."""
        B, S, Din = x.shape
        w, b, neck, nw, nb = lin
        E = w.shape[0]
        T = S if gather is None else int(gather.shape[1])
        out = torch.empty((B, E, T) if channel_first else (B, T, E), device=x.device, dtype=torch.float32)
        arr = (_lib.FeatureTerm * max(1, len(terms)))(*terms)
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        gm = None if gather_mask is None else gather_mask.to(torch.uint8).contiguous()
        with eng.lock:
            _lib.check(_lib.lib().fdx_features_forward_svs(eng.h, _lib.ptr(x), B, S, 0, T, Din, E, _lib.ptr(w), _lib.ptr(b), neck,
                                                           _lib.ptr(nw) if neck else None, _lib.ptr(nb) if neck else None,
                                                           _lib.ptr(gather) if gather is not None else None, _lib.ptr(gm), arr, len(terms), act,
                                                           _lib.ptr(m), int(channel_first), _lib.ptr(out), _lib.stream_ptr(x.device)), eng.h)
        return out

    @staticmethod
    def _lin(layer):
        return layer.weight.detach().to(torch.float32).contiguous(), layer.bias.detach().to(torch.float32).contiguous(), 0, None, None

    @torch.no_grad()
    def forward_features(self, speakers, contents, contents_lens, contents_max_len, pitch_shift=None, phones2mel=None, energy=None,
                         channel_first: bool = False):
        """core.py:55-115.  `channel_first=True` returns features as [B, hidden, T] (what the generator consumes)."""
        if not isinstance(self.text_encoder, NaiveProjectionEncoder) or self.text_encoder.use_embedding:
            raise NotImplementedError("only the NaiveProjectionEncoder (Linear) text encoder is fused")
        if contents_lens is None:
            raise ValueError("contents_lens is required (the reference multiplies by 1 - src_masks unconditionally, core.py:110)")
        _lib.require_gpu(contents, "contents")
        src_masks = self.get_mask_from_lengths(contents_lens, contents_max_len)
        B, T, _ = contents.shape
        p2m = None
        if phones2mel is not None:        # SVS duration gather (core.py:72-78): frame t <- text frame phones2mel[b][t], * (1 - src_mask)
            p2m = phones2mel.to(device=contents.device, dtype=torch.int64).contiguous()
            if p2m.ndim != 2 or p2m.shape[0] != B or tuple(src_masks.shape) != tuple(p2m.shape):
                raise ValueError(f"phones2mel {tuple(phones2mel.shape)} does not match the mask {tuple(src_masks.shape)}")
            if p2m.numel() and (int(p2m.min()) < 0 or int(p2m.max()) >= T):
                raise RuntimeError("index out of range in phones2mel")
            T = int(p2m.shape[1])
        E = self.text_encoder.output_size
        keep, terms = [], []
        if speakers.ndim in (2, 3) and torch.is_floating_point(speakers):
            v = speakers.to(torch.float32).contiguous()
            if v.shape[-1] != E or (v.ndim == 3 and v.shape[1] not in (1, T)):
                raise ValueError(f"speaker embedding {tuple(v.shape)} does not broadcast to [{B}, {T}, {E}]")
            keep.append(v)
            terms.append(_lib.FeatureTerm(_lib.TERM_VECTOR, int(v.ndim == 3 and v.shape[1] == T and T != 1), 0, 0, v.data_ptr(), None, None, 0.0, 0.0))
        else:
            enc = self.speaker_encoder
            ids = speakers.to(torch.int64).reshape(-1).contiguous()
            if ids.numel() != B:
                raise ValueError(f"expected one speaker id per utterance, got {tuple(speakers.shape)}")
            if int(ids.min()) < 0 or int(ids.max()) >= enc.input_size:
                raise IndexError("speaker id out of range")
            tab = enc.embedding.weight.detach().to(torch.float32).contiguous()
            keep += [ids, tab]
            terms.append(_lib.FeatureTerm(_lib.TERM_EMBEDDING, 0, 0, 0, ids.data_ptr(), tab.data_ptr(), None, 0.0, 0.0))
        if pitch_shift is not None and hasattr(self, "pitch_shift_encoder"):
            terms.append(DiffSinger._scalar_term(self.pitch_shift_encoder, pitch_shift, B, T, keep))
        if energy is not None and hasattr(self, "energy_encoder"):
            terms.append(DiffSinger._scalar_term(self.energy_encoder, energy, B, T, keep))
        eng = self._engine(contents.device)
        x = contents.to(torch.float32).contiguous()
        f = self._launch(eng, x, self.text_encoder.linear_params(), terms, _lib.ACT_NONE, None, False, p2m, src_masks if p2m is not None else None)
        f = self._launch(eng, f, self._lin(self.feature_fuser[0]), [], _lib.ACT_SILU, None, False)
        f = self._launch(eng, f, self._lin(self.feature_fuser[2]), [], _lib.ACT_SILU, src_masks, channel_first)
        del keep
        return dict(features=f, src_masks=src_masks)

    @torch.no_grad()
    def forward(self, speakers, contents, contents_lens, contents_max_len, pitches=None, pitch_shift=None, phones2mel=None,
                energy=None, noises=None):
        """core.py:117-141: pitches [B, T, 1] -> waveform [B, 1, T * hop_length]."""
        f = self.forward_features(speakers, contents, contents_lens, contents_max_len, pitch_shift=pitch_shift, phones2mel=phones2mel,
                                  energy=energy, channel_first=True)
        if self.encoder_type == "RefineGAN":
            return self.encoder(f["features"], pitches.transpose(1, 2), noises=noises)
        rand_ini, src_noise = noises if noises is not None else (None, None)   # (rand_ini [B,9], src_noise [B,L,9])
        return self.encoder(f["features"], pitches[:, :, 0], rand_ini=rand_ini, src_noise=src_noise)
