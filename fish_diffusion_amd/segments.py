"""The data formats and the caller loop either side of the hot path:

  * `.npy` feature dictionaries written by tools/preprocessing/extract_features.py:107-172 (`load_sample` / `save_sample`);
  * silence slicing of a long recording, fish_diffusion/utils/audio.py:112-167 (`slice_audio`; `librosa.effects.split`
    -- librosa 0.9.1, third-party, not vendored and absent here -- restated from its published algorithm: that one
    function is unpinned, the chunking logic around it follows the reference line by line);
  * the per-segment loop of SVCInference.inference, tools/diffusion/inference.py:336-376: every segment through
    front end -> sampler -> vocoder, pasted at its own offset (`convert_segments` batches the segments instead of running
    them one by one; `stitch` is the paste `generated_audio[start : start + len(wav)] = wav[:max_len]`);
  * `sf.write(output_path, generated_audio, sr)` (:383-387): a RIFF/WAVE PCM_16 file, soundfile's default subtype for
    `.wav` (`write_wav`; soundfile is absent here, the 44-byte header is written directly).

Host plumbing only: all arithmetic on the waveform/mel path stays in libfishdx.so.
"""
from __future__ import annotations

import math
import os
import struct
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import pipeline


# ------------------------------------------------------------------------------------------------ .npy feature dictionaries
SAMPLE_KEYS = ("path", "audio", "sampling_rate", "time_stretch", "mel", "contents", "phones2mel", "pitches", "key_shift", "energy")


def load_sample(path: str) -> dict:
    """One `*.data.npy` of the reference's preprocessing: a pickled dict of numpy arrays / scalars (np.save(save_path, sample),
    extract_features.py:172; read back by datasets/naive.py with allow_pickle=True)."""
    obj = np.load(path, allow_pickle=True)
    sample = obj.item() if isinstance(obj, np.ndarray) and obj.dtype == object else obj
    if not isinstance(sample, dict):
        raise ValueError(f"{path}: not a feature dictionary")
    return sample


def save_sample(path: str, sample: dict) -> None:
    np.save(path, sample, allow_pickle=True)


def sample_to_device(sample: dict, device: torch.device):
    """(contents [Din, S], pitches [T], mel [M, T] or None) as fp32 device tensors, layouts as stored by the reference."""
    c = torch.as_tensor(np.asarray(sample["contents"]), dtype=torch.float32, device=device)
    p = torch.as_tensor(np.asarray(sample["pitches"]), dtype=torch.float32, device=device).reshape(-1)
    m = sample.get("mel")
    m = None if m is None else torch.as_tensor(np.asarray(m), dtype=torch.float32, device=device)
    return c, p, m


# ------------------------------------------------------------------------------------------------ silence slicing
def _nonsilent_frames(y: np.ndarray, frame_length: int, hop_length: int, top_db: float) -> np.ndarray:
    """librosa.effects._signal_to_frame_nonsilent (0.9.1): centred, reflect-padded frame RMS; power in dB relative to its
    maximum; frames above -top_db."""
    pad = frame_length // 2
    yp = np.pad(np.asarray(y, dtype=np.float32), pad, mode="reflect")
    frames = np.lib.stride_tricks.sliding_window_view(yp, frame_length)[::hop_length]
    mse = np.mean(np.abs(frames) ** 2, axis=-1)
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(np.maximum(1e-10, mse.max()))
    return db > -top_db


def split_nonsilent(y: np.ndarray, top_db: float = 60, frame_length: int = 2048, hop_length: int = 512) -> np.ndarray:
    """librosa.effects.split: [n, 2] sample intervals of the non-silent runs."""
    non_silent = _nonsilent_frames(y, frame_length, hop_length, top_db)
    edges = [np.flatnonzero(np.diff(non_silent.astype(int))) + 1]
    if non_silent[0]:
        edges.insert(0, np.array([0]))
    if non_silent[-1]:
        edges.append(np.array([len(non_silent)]))
    edges = np.minimum(np.concatenate(edges) * hop_length, len(y))
    return edges.reshape((-1, 2))


def slice_audio(audio: np.ndarray, rate: int, max_duration: float = 30.0, top_db: int = 60, frame_length: int = 2048,
                hop_length: int = 512, min_silence_duration: float = 0) -> Iterable[Tuple[int, int]]:
    """fish_diffusion/utils/audio.py:112-167."""
    intervals = [tuple(int(v) for v in iv) for iv in split_nonsilent(audio, top_db, frame_length, hop_length)]
    if min_silence_duration > 0:                         # merge intervals that are too close (:138-151)
        merged: List[Tuple[int, int]] = []
        for start, end in intervals:
            if merged and merged[-1][1] + min_silence_duration * rate >= start:
                merged[-1] = (merged[-1][0], end)
            else:
                merged.append((start, end))
        intervals = merged
    for start, end in intervals:
        if end - start <= rate * max_duration:
            if end - start <= rate * 0.1:                # too short, unlikely to be vocal (:155-157)
                continue
            yield start, end
            continue
        n_chunks = math.ceil((end - start) / (max_duration * rate))
        chunk_size = math.ceil((end - start) / n_chunks)
        for i in range(start, end, chunk_size):
            yield i, i + chunk_size                      # (the last chunk may run past `end`, as in the reference :166-167)


# ------------------------------------------------------------------------------------------------ paste + file output
def stitch(total_len: int, pieces: Sequence[Tuple[int, torch.Tensor]], device=None) -> torch.Tensor:
    """inference.py:336,375-376: zeros_like(audio), then `out[start : start + len(wav)] = wav[: total_len - start]` in order."""
    if device is None:
        device = pieces[0][1].device if pieces else torch.device("cpu")
    out = torch.zeros(int(total_len), dtype=torch.float32, device=device)
    for start, wav in pieces:
        wav = wav.reshape(-1)[: max(0, total_len - start)]
        out[start:start + wav.numel()] = wav
    return out


def write_wav(path: str, audio, sr: int) -> None:
    """sf.write(path, audio, sr) for a `.wav` path: mono/[N, C] float data -> PCM_16 (libsndfile: lrint(x * 32767); samples
    outside [-1, 1] are clipped here, where libsndfile's default conversion would wrap)."""
    a = audio.detach().cpu().numpy() if torch.is_tensor(audio) else np.asarray(audio)
    a = a.astype(np.float32, copy=False)
    if a.ndim == 1:
        a = a[:, None]
    pcm = np.clip(np.rint(a * 32767.0), -32768, 32767).astype("<i2")
    n, ch = pcm.shape
    d = os.path.dirname(path)
    if d and not os.path.exists(d):
        os.makedirs(d)                                   # inference.py:384-385
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + n * ch * 2) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, int(sr), int(sr) * ch * 2, ch * 2, 16)
                + b"data" + struct.pack("<I", n * ch * 2))
        f.write(pcm.tobytes())


# ------------------------------------------------------------------------------------------------ the caller loop, batched
@torch.no_grad()
def convert_segments(model, vocoder, total_len: int, segments: Sequence[Tuple[int, int]], contents: Sequence[torch.Tensor],
                     pitches: Sequence[torch.Tensor], speakers, *, hop: int = 512, contents_channel_first: bool = True,
                     pitch_adjust: float = 0.0, max_batch: int = 8, sampler_interval: Optional[int] = None,
                     noise_predictor: Optional[str] = None, rank: int = 0, world: int = 1, **synth_kwargs) -> torch.Tensor:
    """SVCInference.inference's segment loop (inference.py:336-376) with the segments batched through the sampler:
    per segment `mel_len = (end - start) // 512` (:104), front end with the extractor's frames nearest-expanded inside the
    launch (:108-114), pitch * 2^(pitch_adjust/12) (:111), an all-unvoiced segment yields silence (:108-109 `return zeros`),
    then `pipeline.synthesize` (length-bucketed masked sampler batches, vocoder per segment) and the paste.
    contents[i]: the extractor's output for segment i, `[Din, S_i]` (or `[S_i, Din]`); pitches[i]: `[P_i]` f0 in Hz at any
    frame count (expanded to mel_len); speakers: what `DiffSinger.forward_features` takes for ONE utterance (id tensor [1] or
    a float mix [1, E])."""
    feats, f0s, keep = [], [], []
    for i, (start, end) in enumerate(segments):
        # the reference slices audio[start:end], which clips at the end of the audio, and takes mel_len from the CLIPPED
        # segment (inference.py:104); slice_audio's last chunk may run past the interval (and the audio) end
        mel_len = (min(end, total_len) - start) // hop
        if mel_len <= 0:
            continue
        p = pitches[i].to(torch.float32).reshape(1, -1)
        if bool((p == 0).all()):
            continue                                     # silence stays silence
        p = p * (2.0 ** (pitch_adjust / 12.0))
        c = contents[i].to(torch.float32)[None]
        lens = torch.tensor([mel_len], device=c.device)
        f = model.forward_features(speakers=speakers, contents=c, contents_lens=lens, contents_max_len=mel_len, pitches=p,
                                   contents_channel_first=contents_channel_first, expand_to=mel_len)["features"][0]
        from .diffsinger import repeat_expand
        feats.append(f)
        f0s.append(repeat_expand(p[0], mel_len) if p.shape[1] != mel_len else p[0])
        keep.append(i)
    pieces = []
    if feats:
        res = pipeline.synthesize(model.diffusion, vocoder, feats, f0s, max_batch=max_batch, sampler_interval=sampler_interval,
                                  noise_predictor=noise_predictor, rank=rank, world=world, **synth_kwargs)
        by_idx = {j: wav for j, _, wav in res}
        pieces = [(segments[keep[j]][0], by_idx[j]) for j in sorted(by_idx)]   # paste in segment order, like the loop
    dev = feats[0].device if feats else None
    return stitch(total_len, pieces, device=dev)
