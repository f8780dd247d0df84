"""Batched, utterance-sharded inference: features -> sampler -> vocoder for MANY utterances (BASELINE configs[3]:
"end-to-end denoise+vocoder, batch=64 utterances sharded over 8 MI355X").

What the reference does one utterance at a time in `SVCInference.forward` (tools/diffusion/inference.py:86-162, batch
dimension always 1) and shards across GPUs by rank-strided file lists (tools/preprocessing/extract_features.py:262-322),
this module does for a list of utterances per process:

  * `dist.shard_utterances`: longest-first round-robin over ranks (no collective);
  * micro-batches of similar length, padded to the longest member rounded UP to a 64-frame bucket (the kernels' column tile: the
    padding costs no extra tiles), with `x_masks` / `cond_masks`
    (`DiffSinger.get_mask_from_lengths`, diffsinger.py:42-55) so that padding never leaks into valid frames
    (masked conditioner, masked denoiser input/output: wavenet.py:217-221,233-234);
  * one `GaussianDiffusion` call per micro-batch; the sampler body is a recorded hipGraph per (batch size, padded length), and the
    buckets keep the number of distinct geometries a ragged stream produces to a few dozen, all of which stay cached: a serving
    loop re-captures nothing in steady state -- the reference's own batched +
    masked semantics (what its validation loop runs); frames within the receptive field of an utterance's end see the
    masked tail's activations instead of zero padding, exactly as in the reference, so they are not bit-identical to a
    one-by-one run;
  * the vocoder runs per utterance on the unpadded mel (the reference's Generator has no length masks: a padded batch
    would leak the padding into the last few hundred samples); the utterances of a micro-batch go through it on a few HIP
    streams side by side (`vocoder_lanes`), each pass still the batch-1 pass, so every waveform is what a call alone gives;
  * results cut back to each utterance's own length.

Host-side plumbing only -- all arithmetic is in libfishdx.so.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import dist as fdist


# Relative throughput of one sampler launch sequence at batch B against batch 1 (measured: profiles/README.md, 10 s utterances,
# 100-step UniPC: 113 / 121 / 128 / 143 audio-s/s at B = 1 / 2 / 4 / 8): what padding a short utterance up to a longer one costs
# has to be weighed against how much better a fuller batch fills the chip.
_BATCH_EFF = ((1, 1.00), (2, 1.07), (4, 1.13), (8, 1.27), (16, 1.38))


def _eff(b: int) -> float:
    if b <= 1:
        return 1.0
    for (b0, e0), (b1, e1) in zip(_BATCH_EFF, _BATCH_EFF[1:]):
        if b <= b1:
            return e0 + (e1 - e0) * (b - b0) / (b1 - b0)
    return _BATCH_EFF[-1][1]


def make_batches(lengths: Sequence[int], max_batch: int, max_pad_ratio: Optional[float] = None, padding_free: bool = False) -> List[List[int]]:
    """Group utterance indices (given in ANY order) into micro-batches of at most `max_batch`, each padded to its longest
    member.  Utterances are sorted longest-first and cut into consecutive groups; the cut minimises the modelled time
    sum(len(group) * longest(group) / eff(len(group))) by dynamic programming (padding wastes frames, small batches waste the
    chip).  `max_pad_ratio`, when given, additionally forbids padding any member by more than that fraction of the group's
    longest (the plain greedy rule).  `padding_free`: the batch runs in exact-ragged mode, where padding costs no arithmetic (the
    group's cost is the sum of its members' own lengths): fuller batches always win."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    n = len(order)
    if n == 0:
        return []
    L = [int(lengths[i]) for i in order]
    INF = float("inf")
    best = [0.0] + [INF] * n          # best[j]: cost of the first j utterances
    cut = [0] * (n + 1)
    for j in range(1, n + 1):
        for i in range(max(0, j - max_batch), j):          # group = order[i:j], longest = L[i]
            if max_pad_ratio is not None and L[j - 1] < (1.0 - max_pad_ratio) * L[i]:
                continue
            c = best[i] + (sum(L[i:j]) if padding_free else (j - i) * L[i]) / _eff(j - i)
            if c < best[j] - 1e-9:
                best[j], cut[j] = c, i
    batches: List[List[int]] = []
    j = n
    while j > 0:
        batches.append(order[cut[j]:j])
        j = cut[j]
    return batches[::-1]


@torch.no_grad()
def _lane_count(requested: int, B: int, samples: int, dev) -> int:
    """Vocoder lanes for one micro-batch: never more than its utterances, and never more than free device memory holds -- a lane's
    workspace is every stage's activations of one utterance at once (~2 kB per output sample for config_v1: 0.9 GB per 10 s), cached on
    the Generator until `release_lanes()`.  Keeps half of what is free for the denoiser's next micro-batch."""
    n = min(requested, B)
    try:
        free, _ = torch.cuda.mem_get_info(dev)
        n = min(n, int(free // 2 // max(1, 2048 * samples)))
    except Exception:        # no such query on this build: the caller's number stands
        pass
    return max(0, n)


def synthesize(diffusion, vocoder, features: Sequence[torch.Tensor], f0s: Sequence[torch.Tensor], *, max_batch: int = 8,
               sampler_interval: Optional[int] = None, noise_predictor: Optional[str] = None, rank: int = 0, world: int = 1,
               mel_scale: Optional[float] = None, x_init_fn: Optional[Callable] = None,
               source_noise_fn: Optional[Callable] = None, bucket: int = 64, exact: Optional[bool] = None,
               on_error: str = "raise", failures: Optional[list] = None,
               vocoder_lanes: Optional[int] = None) -> List[Tuple[int, torch.Tensor, torch.Tensor]]:
    """features[i]: [T_i, E] device tensors; f0s[i]: [T_i].  Returns [(index, mel [T_i, M], wav [T_i * hop])] for the
    utterances this rank owns.  `x_init_fn(idx_list, M, T)` / `source_noise_fn(idx_list, L)` let tests inject the random
    draws (initial x_T; (rand_ini, src_noise)) -- by default they are drawn on the device.  `bucket`: every micro-batch is padded
    to a multiple of this many frames (0 / 1 = pad to the longest member only).
    `exact` (default: True for the three HIP denoisers in fp32 storage, and the WaveNet in fp16x3): padded batches run in the library's EXACT-RAGGED mode
    -- every utterance's result is what a batch-1 run of it alone gives (the reference's one-segment-at-a-time loop): bit for bit in
    fp32 storage, to fp32 rounding in the opt-in fp16x3 storage (a long row may run the 128-wide fp16-split tiles where the short item alone
    runs the 64 x 64 fp16-split tiles: both fp32-class, not bit-identical to each other) -- and padding costs no arithmetic.  Whether the
    mode exists is a static property of (denoiser, storage): decided ONCE, before batching (bf16 storage has no exact-mask kernels: an
    explicit `exact=True` raises NotImplementedError from the library, the default picks the reference's masked batches there; the
    ConvNext and transformer denoisers run exact since round 5 -- attention per item, positions restarting at every item).  False: the
    reference's own padded-batch semantics with x_masks /
    cond_masks (the masked tail stays alive inside the receptive field: the last ~75 frames of every padded item differ slightly
    from a run alone).
    `vocoder_lanes` (default 4, env FDX_VOC_LANES): the utterances of a micro-batch go through the generator on this many HIP streams side
    by side (`Generator.lanes`): every pass is still the batch-1 pass on the unpadded mel -- bit-identical waveforms -- but the small-grid
    stages of different utterances fill each other's idle CUs.  0 / 1: one after the other on the caller's stream.  Each lane keeps a
    vocoder workspace of its own (~0.9 GB per 10 s of audio) cached on the Generator: the count is capped by the micro-batch size and by
    free device memory (`_lane_count`), `Generator.release_lanes()` frees them.
    `on_error`: "raise" (default) lets the first exception out, as a plain loop would.  "isolate" is the reference's `safe_process`
    (tools/preprocessing/extract_features.py:175-217: one bad file is logged and the worker carries on): an utterance that fails validation
    is skipped; a micro-batch that raises is re-run one member at a time so that a bad member does not cost its batch-mates; whatever
    still fails alone is appended to `failures` as `(index, "ExcType: message")` and left out of the result.  The ids are this rank's;
    `dist.gather_failed` collects every rank's (the reference counts `failed` per worker, :298-305)."""
    if on_error not in ("raise", "isolate"):
        raise ValueError('on_error must be "raise" or "isolate"')
    if len(features) != len(f0s):
        raise ValueError("features and f0s must have the same length")
    if failures is None:
        failures = []
    lengths = [int(f.shape[0]) if hasattr(f, "shape") and len(f.shape) >= 1 else 0 for f in features]
    mine = fdist.shard_utterances(lengths, rank, world)
    if not mine:
        return []

    def invalid(i) -> Optional[str]:
        f, p = features[i], f0s[i]
        if not torch.is_tensor(f) or f.dim() != 2 or f.shape[0] == 0:
            return f"features must be a non-empty [T, E] tensor, got {tuple(f.shape) if torch.is_tensor(f) else type(f).__name__}"
        if not torch.is_tensor(p) or p.dim() != 1 or p.shape[0] != f.shape[0]:
            return f"f0 must be [T = {f.shape[0]}], got {tuple(p.shape) if torch.is_tensor(p) else type(p).__name__}"
        if width is not None and f.shape[1] != width:
            return f"feature width {f.shape[1]} differs from the job's {width}"
        return None

    # the job's feature width: the first well-formed utterance's (a malformed first entry must not decide it, or take the check down)
    width = next((int(features[i].shape[1]) for i in mine if torch.is_tensor(features[i]) and features[i].dim() == 2 and features[i].shape[0] > 0), None)

    if on_error == "isolate":
        ok = []
        for i in mine:
            why = invalid(i)
            if why is None:
                ok.append(i)
            else:
                failures.append((i, "ValueError: " + why))
        mine = ok
        if not mine:
            return []
    gen = vocoder.model if hasattr(vocoder, "model") else vocoder
    hop = gen.h["hop_size"]
    if mel_scale is None:   # nsf_hifigan.py:79-80: a log10 mel is rescaled to natural log
        mel_scale = 2.30259 if getattr(vocoder, "use_natural_log", True) is False else 1.0
    dev = features[mine[0]].device
    if exact is None:
        den = getattr(diffusion, "denoise_fn", None)
        kind = getattr(den, "_KIND", "")
        exact = (kind == "wavenet" and getattr(den, "storage", "fp32") in ("fp32", "fp16x3")) or kind in ("convnext", "tfdec")
    if vocoder_lanes is None:
        vocoder_lanes = int(os.environ.get("FDX_VOC_LANES", "4"))
    use_lanes = vocoder_lanes > 1 and hasattr(gen, "lanes")

    def run_group(idx: List[int]) -> List[Tuple[int, torch.Tensor, torch.Tensor]]:
        T = max(lengths[i] for i in idx)
        if bucket and bucket > 1:
            T = (T + bucket - 1) // bucket * bucket
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
        if ragged and exact:
            kw["lengths"] = [lengths[i] for i in idx]
        elif ragged:
            kw["x_masks"] = kw["cond_masks"] = masks
        mel = diffusion(feat, sampler_interval=sampler_interval, noise_predictor=noise_predictor, **kw)     # [B, T, M]
        res = []
        lanes = gen.lanes(dev, _lane_count(vocoder_lanes, B, T * hop, dev)) if use_lanes and B > 1 else []
        if len(lanes) < 2:
            lanes = []
        cur = torch.cuda.current_stream(dev) if lanes else None
        try:
            for b, i in enumerate(idx):
                n = lengths[i]
                vkw = {}
                if source_noise_fn is not None:
                    vkw["rand_ini"], vkw["src_noise"] = source_noise_fn([i], n * hop)
                m_i = mel[b, :n]
                if lanes:
                    eng, side = lanes[b % len(lanes)]
                    side.wait_stream(cur)                      # the mel, f0 and any injected noise are the caller stream's work
                    with torch.cuda.stream(side):
                        wav = gen(m_i.T[None].contiguous(), f0[b:b + 1, :n].contiguous(), mel_scale=mel_scale, engine=eng, **vkw)[0, 0]
                    wav.record_stream(cur)                     # allocated on the lane's stream, consumed on the caller's
                else:
                    wav = gen(m_i.T[None].contiguous(), f0[b:b + 1, :n].contiguous(), mel_scale=mel_scale, **vkw)[0, 0]   # [n*hop]
                res.append((i, m_i, wav))
        finally:
            for _, side in lanes:                              # join: whatever the caller does next sees finished waveforms
                cur.wait_stream(side)
        return res

    def guarded(idx: List[int]) -> List[Tuple[int, torch.Tensor, torch.Tensor]]:
        try:
            return run_group(idx)
        except Exception as e:   # noqa: BLE001 -- safe_process catches everything a single utterance can throw
            if on_error == "raise":
                raise
            if len(idx) == 1:
                failures.append((idx[0], f"{type(e).__name__}: {e}"))
                return []
            res = []
            for i in idx:        # the batch failed as a whole: find the member(s) that fail alone, keep the others
                res += guarded([i])
            return res

    out = []
    for group in make_batches([lengths[i] for i in mine], max_batch, padding_free=bool(exact)):
        out += guarded([mine[g] for g in group])
    return out
