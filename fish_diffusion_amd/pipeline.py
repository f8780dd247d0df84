"""Batched, utterance-sharded inference: features -> sampler -> vocoder for MANY utterances (BASELINE configs[3]:
"end-to-end denoise+vocoder, batch=64 utterances sharded over 8 MI355X").

What the reference does one utterance at a time in `SVCInference.forward` (tools/diffusion/inference.py:86-162, batch
dimension always 1) and shards across GPUs by rank-strided file lists (tools/preprocessing/extract_features.py:262-322),
this module does for a list of utterances per process:

  * `dist.shard_utterances`: longest-first round-robin over ranks (no collective);
  * micro-batches of similar length, padded to the longest member, with `x_masks` / `cond_masks`
    (`DiffSinger.get_mask_from_lengths`, diffsinger.py:42-55) so that padding never leaks into valid frames
    (masked conditioner, masked denoiser input/output: wavenet.py:217-221,233-234);
  * one `GaussianDiffusion` call (a recorded hipGraph per geometry) per micro-batch -- the reference's own batched +
    masked semantics (what its validation loop runs); frames within the receptive field of an utterance's end see the
    masked tail's activations instead of zero padding, exactly as in the reference, so they are not bit-identical to a
    one-by-one run;
  * the vocoder runs per utterance on the unpadded mel (the reference's Generator has no length masks: a padded batch
    would leak the padding into the last few hundred samples), batch 1 is already efficient there;
  * results cut back to each utterance's own length.

Host-side plumbing only -- all arithmetic is in libfishdx.so.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import dist as fdist


def make_batches(lengths: Sequence[int], max_batch: int, max_pad_ratio: float = 0.25) -> List[List[int]]:
    """Group utterance indices (given in ANY order) into micro-batches: walk them longest-first and close a batch when it
    is full or when the next utterance would be padded by more than `max_pad_ratio` of the batch's longest member."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    batches: List[List[int]] = []
    cur: List[int] = []
    for i in order:
        if cur and (len(cur) >= max_batch or lengths[i] < (1.0 - max_pad_ratio) * lengths[cur[0]]):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


@torch.no_grad()
def synthesize(diffusion, vocoder, features: Sequence[torch.Tensor], f0s: Sequence[torch.Tensor], *, max_batch: int = 8,
               sampler_interval: Optional[int] = None, noise_predictor: Optional[str] = None, rank: int = 0, world: int = 1,
               mel_scale: Optional[float] = None, x_init_fn: Optional[Callable] = None,
               source_noise_fn: Optional[Callable] = None) -> List[Tuple[int, torch.Tensor, torch.Tensor]]:
    """features[i]: [T_i, E] device tensors; f0s[i]: [T_i].  Returns [(index, mel [T_i, M], wav [T_i * hop])] for the
    utterances this rank owns.  `x_init_fn(idx_list, M, T)` / `source_noise_fn(idx_list, L)` let tests inject the random
    draws (initial x_T; (rand_ini, src_noise)) -- by default they are drawn on the device."""
    if len(features) != len(f0s):
        raise ValueError("features and f0s must have the same length")
    lengths = [int(f.shape[0]) for f in features]
    mine = fdist.shard_utterances(lengths, rank, world)
    if not mine:
        return []
    gen = vocoder.model if hasattr(vocoder, "model") else vocoder
    hop = gen.h["hop_size"]
    if mel_scale is None:   # nsf_hifigan.py:79-80: a log10 mel is rescaled to natural log
        mel_scale = 2.30259 if getattr(vocoder, "use_natural_log", True) is False else 1.0
    dev = features[mine[0]].device
    out = []
    for group in make_batches([lengths[i] for i in mine], max_batch):
        idx = [mine[g] for g in group]
        T = max(lengths[i] for i in idx)
        B = len(idx)
        feat = torch.zeros((B, T, features[idx[0]].shape[1]), device=dev, dtype=torch.float32)
        f0 = torch.zeros((B, T), device=dev, dtype=torch.float32)
        lens = torch.tensor([lengths[i] for i in idx], device=dev)
        for b, i in enumerate(idx):
            feat[b, :lengths[i]] = features[i]
            f0[b, :lengths[i]] = f0s[i]
        masks = torch.arange(T, device=dev)[None, :] >= lens[:, None]          # True = padding
        ragged = bool(masks.any())
        kw = {}
        if x_init_fn is not None:
            kw["x_init"] = x_init_fn(idx, diffusion.mel_bins, T)
        mel = diffusion(feat, sampler_interval=sampler_interval, noise_predictor=noise_predictor,
                        x_masks=masks if ragged else None, cond_masks=masks if ragged else None, **kw)     # [B, T, M]
        for b, i in enumerate(idx):
            n = lengths[i]
            vkw = {}
            if source_noise_fn is not None:
                vkw["rand_ini"], vkw["src_noise"] = source_noise_fn([i], n * hop)
            m_i = mel[b, :n]
            wav = gen(m_i.T[None].contiguous(), f0[b:b + 1, :n].contiguous(), mel_scale=mel_scale, **vkw)[0, 0]   # [n*hop]
            out.append((i, m_i, wav))
    return out
