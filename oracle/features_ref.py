"""CPU oracle for the condition front end.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Functional restatement (state-dict in, features out) of
  fish_diffusion/archs/diffsinger/diffsinger.py:42-55 (``get_mask_from_lengths``), :57-134 (``forward_features``)
for the encoders the SVC configs build (configs/_base_/archs/diff_svc_v2.py:41-58):
  fish_diffusion/modules/encoders/naive_projection.py:6-60 (Linear / Embedding), utils/pitch.py:12-22 (``pitch_to_scale``).
Pinned by oracle/make_golden.py against the reference's own ``forward_features`` source (extracted from the file and
executed unmodified on real ``NaiveProjectionEncoder`` instances; the enclosing module cannot be imported here because
it pulls lightning / loralib / wandb at import time).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

F0_MIN, F0_MAX = 50.0, 1100.0  # utils/pitch.py:6-7


def pitch_to_scale(f0: torch.Tensor, f0_min=F0_MIN, f0_max=F0_MAX) -> torch.Tensor:
    """utils/pitch.py:12-22."""
    s = (f0 - f0_min) / (f0_max - f0_min)
    s = s.clone()
    s[s < 0] = 0
    s[s > 1] = 1
    return s.unsqueeze(-1) if f0.ndim == 2 else s


def repeat_expand(content: torch.Tensor, target_len: int) -> torch.Tensor:
    """utils/tensor.py:7-43 (mode "nearest"): [S] / [C, S] / [B, C, S] -> target_len frames."""
    nd = content.ndim
    x = content[None, None] if nd == 1 else content[None] if nd == 2 else content
    y = torch.nn.functional.interpolate(x, size=target_len, mode="nearest")
    return y[0, 0] if nd == 1 else y[0] if nd == 2 else y


def mask_from_lengths(lengths: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """diffsinger.py:42-55: True = padding."""
    if max_len is None:
        max_len = int(lengths.max())
    ids = torch.arange(0, int(max_len)).unsqueeze(0).expand(lengths.shape[0], -1)
    return ids >= lengths.unsqueeze(1).expand(-1, int(max_len))


def _projection(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """NaiveProjectionEncoder.forward without preprocessing (naive_projection.py:34-44,57-60): nn.Linear, or -- use_neck --
    nn.Sequential(Linear(in, neck), Linear(neck, out)) with state-dict keys projection.0.* / projection.1.*."""
    if f"{prefix}.projection.0.weight" in sd:
        h = F.linear(x, sd[f"{prefix}.projection.0.weight"], sd[f"{prefix}.projection.0.bias"])
        return F.linear(h, sd[f"{prefix}.projection.1.weight"], sd[f"{prefix}.projection.1.bias"])
    return F.linear(x, sd[f"{prefix}.projection.weight"], sd[f"{prefix}.projection.bias"])


def _has(sd: SD, prefix: str) -> bool:
    return f"{prefix}.projection.weight" in sd or f"{prefix}.projection.0.weight" in sd


def forward_features(sd: SD, contents, speakers=None, pitches=None, pitch_shift=None, energy=None, mel_lens=None,
                     mel_max_len=None, phones2mel=None) -> dict:
    """diffsinger.py:57-134.  Keys of ``sd`` are the DiffSinger state-dict names:
    text_encoder.projection.{weight,bias} (use_neck: projection.{0,1}.*), speaker_encoder.embedding.weight,
    {pitch,pitch_shift,energy}_encoder.projection.*"""
    mel_masks = mask_from_lengths(mel_lens, mel_max_len) if mel_lens is not None else None
    features = _projection(sd, "text_encoder", contents)
    if phones2mel is not None:                                  # :85-90, the SVS duration gather
        idx = phones2mel.unsqueeze(-1).repeat([1, 1, features.shape[-1]]).long()
        features = torch.gather(features, 1, idx) * (1 - mel_masks[:, :, None].float())
    emb = None
    if speakers is not None and speakers.ndim in (2, 3) and torch.is_floating_point(speakers):
        emb = speakers
    elif speakers is not None and "speaker_encoder.embedding.weight" in sd:
        emb = F.embedding(speakers, sd["speaker_encoder.embedding.weight"])
    if emb is not None and emb.ndim == 2:
        emb = emb[:, None, :]
    if emb is not None:
        features = features + emb
    if _has(sd, "pitch_encoder"):
        features = features + _projection(sd, "pitch_encoder", pitch_to_scale(pitches))
    if pitch_shift is not None and _has(sd, "pitch_shift_encoder"):
        e = _projection(sd, "pitch_shift_encoder", pitch_shift)
        features = features + (e[:, None, :] if e.ndim == 2 else e)
    if energy is not None and _has(sd, "energy_encoder"):
        e = _projection(sd, "energy_encoder", energy)
        features = features + (e[:, None, :] if e.ndim == 2 else e)
    return dict(features=features, x_masks=mel_masks, x_lens=mel_lens, cond_masks=mel_masks)


def seeded_svs_frontend_state(seed: int, content_dim=64, hidden=96, n_speakers=10, neck=8) -> SD:
    """A front end whose text, pitch and energy encoders use the bottleneck variant (use_neck=True, neck_size=neck)."""
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f):
        bound = (6.0 / (out_f + in_f)) ** 0.5
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound, (torch.rand(out_f, generator=g) * 2 - 1) * 0.05

    sd = {}
    for name, din in (("text_encoder", content_dim), ("pitch_encoder", 1), ("energy_encoder", 1)):
        sd[f"{name}.projection.0.weight"], sd[f"{name}.projection.0.bias"] = lin(neck, din)
        sd[f"{name}.projection.1.weight"], sd[f"{name}.projection.1.bias"] = lin(hidden, neck)
    sd["speaker_encoder.embedding.weight"] = torch.randn(n_speakers, hidden, generator=g) * hidden ** -0.5
    return sd


def seeded_frontend_state(seed: int, content_dim=256, hidden=256, n_speakers=10, pitch_shift=False, energy=False) -> SD:
    """Weights with the reference initialisers' statistics (xavier_uniform_ / N(0, hidden^-0.5), naive_projection.py:48-55);
    biases get small noise instead of the reference's 0 so the bias paths are exercised."""
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f):
        bound = (6.0 / (out_f + in_f)) ** 0.5
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound, (torch.rand(out_f, generator=g) * 2 - 1) * 0.05

    sd = {}
    sd["text_encoder.projection.weight"], sd["text_encoder.projection.bias"] = lin(hidden, content_dim)
    sd["speaker_encoder.embedding.weight"] = torch.randn(n_speakers, hidden, generator=g) * hidden ** -0.5
    sd["pitch_encoder.projection.weight"], sd["pitch_encoder.projection.bias"] = lin(hidden, 1)
    if pitch_shift:
        sd["pitch_shift_encoder.projection.weight"], sd["pitch_shift_encoder.projection.bias"] = lin(hidden, 1)
    if energy:
        sd["energy_encoder.projection.weight"], sd["energy_encoder.projection.bias"] = lin(hidden, 1)
    return sd


# ------------------------------------------------------------------------------------------------ HiFiSinger (hifi_svc_v2)
def hifisinger_features(sd: SD, contents, speakers, contents_lens, contents_max_len, pitch_shift=None, energy=None,
                        phones2mel=None) -> dict:
    """archs/hifisinger/core.py:55-115: text Linear + speaker embedding (+ pitch-shift / energy
    projections), then feature_fuser = Linear, SiLU, Linear, SiLU (:24-29) and `features *= 1 - src_masks` (:109-110).
    Keys: text_encoder.projection.*, speaker_encoder.embedding.weight, {pitch_shift,energy}_encoder.projection.*,
    feature_fuser.{0,2}.{weight,bias}."""
    src_masks = mask_from_lengths(contents_lens, contents_max_len) if contents_lens is not None else None
    features = F.linear(contents, sd["text_encoder.projection.weight"], sd["text_encoder.projection.bias"])
    if phones2mel is not None:                                  # core.py:72-78
        idx = phones2mel.unsqueeze(-1).repeat([1, 1, features.shape[-1]]).long()
        features = torch.gather(features, 1, idx) * (1 - src_masks[:, :, None].float())
    if speakers.ndim in (2, 3) and torch.is_floating_point(speakers):
        emb = speakers
    else:
        emb = F.embedding(speakers, sd["speaker_encoder.embedding.weight"])
    if emb.ndim == 2:
        emb = emb[:, None, :]
    features = features + emb
    if pitch_shift is not None and "pitch_shift_encoder.projection.weight" in sd:
        e = F.linear(pitch_shift, sd["pitch_shift_encoder.projection.weight"], sd["pitch_shift_encoder.projection.bias"])
        features = features + (e[:, None, :] if e.ndim == 2 else e)
    if energy is not None and "energy_encoder.projection.weight" in sd:
        e = F.linear(energy, sd["energy_encoder.projection.weight"], sd["energy_encoder.projection.bias"])
        features = features + (e[:, None, :] if e.ndim == 2 else e)
    features = F.silu(F.linear(features, sd["feature_fuser.0.weight"], sd["feature_fuser.0.bias"]))
    features = F.silu(F.linear(features, sd["feature_fuser.2.weight"], sd["feature_fuser.2.bias"]))
    features = features * (1 - src_masks[:, :, None].float())
    return dict(features=features, src_masks=src_masks)


def seeded_hifisinger_state(seed: int, content_dim=768, hidden=256, n_speakers=10) -> SD:
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f):
        bound = (6.0 / (out_f + in_f)) ** 0.5
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound, (torch.rand(out_f, generator=g) * 2 - 1) * 0.05

    sd = {}
    sd["text_encoder.projection.weight"], sd["text_encoder.projection.bias"] = lin(hidden, content_dim)
    sd["speaker_encoder.embedding.weight"] = torch.randn(n_speakers, hidden, generator=g) * hidden ** -0.5
    sd["pitch_shift_encoder.projection.weight"], sd["pitch_shift_encoder.projection.bias"] = lin(hidden, 1)
    sd["energy_encoder.projection.weight"], sd["energy_encoder.projection.bias"] = lin(hidden, 1)
    sd["feature_fuser.0.weight"], sd["feature_fuser.0.bias"] = lin(hidden, hidden)
    sd["feature_fuser.2.weight"], sd["feature_fuser.2.bias"] = lin(hidden, hidden)
    return sd
