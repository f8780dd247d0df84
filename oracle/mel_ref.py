"""CPU oracle for the STFT/mel front end.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Restates, on CPU:

* ``librosa.filters.mel`` (third-party, NOT vendored in /root/reference; pinned
  ``librosa==0.9.1`` in the reference's pdm.lock) -- called at
  fish_diffusion/utils/pitch_adjustable_mel.py:44-53 with
  ``sr, n_fft, n_mels, fmin, fmax`` and librosa defaults ``htk=False, norm="slaney"``.
  librosa is not installed in this image, so the filterbank itself is "parity
  unpinned" against librosa; it is restated from librosa 0.9.1's published algorithm.
* ``PitchAdjustableMelSpectrogram.__call__`` -- fish_diffusion/utils/pitch_adjustable_mel.py:33-96
* ``dynamic_range_compression``             -- fish_diffusion/utils/audio.py:11-18
* ``NsfHifiGAN.wav2spec`` scalar tail       -- fish_diffusion/modules/vocoders/nsf_hifigan/nsf_hifigan.py:91-107
"""
from __future__ import annotations

import numpy as np
import torch


# ----------------------------------------------------------------------------------------
# librosa 0.9.1 slaney mel scale (librosa/core/convert.py hz_to_mel / mel_to_hz, htk=False)
# ----------------------------------------------------------------------------------------
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    mels = f / _F_SP
    log_t = f >= _MIN_LOG_HZ
    mels = np.where(log_t, _MIN_LOG_MEL + np.log(np.maximum(f, 1e-30) / _MIN_LOG_HZ) / _LOGSTEP, mels)
    return mels


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    freqs = _F_SP * m
    log_t = m >= _MIN_LOG_MEL
    return np.where(log_t, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), freqs)


def slaney_mel_filterbank(*, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, dtype=np.float32):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney') -> [n_mels, 1+n_fft//2]."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    weights = np.zeros((n_mels, n_bins), dtype=dtype)
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


# ----------------------------------------------------------------------------------------
# pitch-adjustable mel spectrogram
# ----------------------------------------------------------------------------------------
def stft_geometry(n_fft, win_size, hop, key_shift=0.0, speed=1.0):
    """pitch_adjustable_mel.py:34-37 -- returns (n_fft_new, win_new, hop_new, pad)."""
    factor = 2 ** (key_shift / 12)
    n_fft_new = int(np.round(n_fft * factor))
    win_new = int(np.round(win_size * factor))
    hop_new = int(np.round(hop * speed))
    pad = int((win_new - hop_new) / 2)
    return n_fft_new, win_new, hop_new, pad


def mel_spectrogram(y: torch.Tensor, *, sample_rate=44100, n_fft=2048, win_size=2048, hop=512,
                    f_min=40, f_max=16000, n_mels=128, key_shift=0.0, speed=1.0) -> torch.Tensor:
    """pitch_adjustable_mel.py:33-96.  y [B, N] float32 -> linear-amplitude mel [B, n_mels, T]."""
    n_fft_new, win_new, hop_new, pad = stft_geometry(n_fft, win_size, hop, key_shift, speed)
    basis = torch.from_numpy(slaney_mel_filterbank(sr=sample_rate, n_fft=n_fft, n_mels=n_mels,
                                                   fmin=f_min, fmax=f_max)).float()
    window = torch.hann_window(win_new)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft_new, hop_length=hop_new, win_length=win_new, window=window,
                      center=False, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    spec = torch.view_as_real(spec)
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    if key_shift != 0:
        size = n_fft // 2 + 1
        resize = spec.size(1)
        if resize < size:
            spec = torch.nn.functional.pad(spec, (0, 0, 0, size - resize))
        spec = spec[:, :size, :] * win_size / win_new
    return torch.matmul(basis, spec)


def wav2spec(wav: torch.Tensor, *, use_natural_log=True, key_shift=0.0, speed=1.0, **mel_kwargs) -> torch.Tensor:
    """nsf_hifigan.py:91-107 (no resampling) + audio.py:11-18.  wav [1, N] -> log-mel [n_mels, T]."""
    mel = mel_spectrogram(wav, key_shift=key_shift, speed=speed, **mel_kwargs)[0]
    mel = torch.log(torch.clamp(mel, min=1e-5) * 1)
    if use_natural_log is False:
        mel = 0.434294 * mel
    return mel
