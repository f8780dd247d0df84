"""`PitchAdjustableMelSpectrogram` on MI355X (fish_diffusion/utils/pitch_adjustable_mel.py:9-96):
reflect pad + Hann + DFT + magnitude + slaney mel filterbank, executed by libfishdx.so (`fdx_mel_forward`)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib


class PitchAdjustableMelSpectrogram:
    def __init__(self, sample_rate=44100, n_fft=2048, win_length=2048, hop_length=512, f_min=40, f_max=16000,
                 n_mels=128, center=False):
        if center:
            raise NotImplementedError("center=True is never used by the reference (pitch_adjustable_mel.py:20)")
        self.sample_rate, self.n_fft, self.win_size, self.hop_length = sample_rate, n_fft, win_length, hop_length
        self.f_min, self.f_max, self.n_mels, self.center = f_min, f_max, n_mels, center
        self._desc = _lib.MelDesc(sample_rate, n_fft, win_length, hop_length, n_mels, float(f_min), float(f_max))
        self._engines: Dict[torch.device, _lib.Handle] = {}

    def filterbank(self) -> torch.Tensor:
        """The slaney mel basis [n_mels, 1 + n_fft//2] (host; librosa.filters.mel semantics)."""
        out = torch.empty((self.n_mels, 1 + self.n_fft // 2), dtype=torch.float32)
        _lib.check(_lib.lib().fdx_mel_filterbank(C.byref(self._desc), C.c_void_p(out.data_ptr())))
        return out

    def num_frames(self, n_samples: int, key_shift=0, speed=1.0) -> int:
        T = C.c_int()
        _lib.check(_lib.lib().fdx_mel_num_frames(C.byref(self._desc), n_samples, float(key_shift), float(speed), C.byref(T)))
        return T.value

    def _engine(self, device: torch.device) -> _lib.Handle:
        device = torch.device("cuda", torch.cuda.current_device() if device.index is None else device.index)
        eng = self._engines.get(device)
        if eng is None:
            eng = _lib.Handle(device)
            _lib.check(_lib.lib().fdx_mel_config(eng.h, C.byref(self._desc)), eng.h)
            self._engines[device] = eng
        return eng

    @torch.no_grad()
    def __call__(self, y: torch.Tensor, key_shift=0, speed=1.0, log_mode=_lib.MEL_LINEAR) -> torch.Tensor:
        """y [B, N] in [-1, 1] -> mel [B, n_mels, T] (linear amplitude unless `log_mode` asks for the fused log)."""
        _lib.require_gpu(y, "PitchAdjustableMelSpectrogram input")
        if y.dim() != 2:
            raise ValueError(f"expected [B, N] audio, got {tuple(y.shape)}")
        y = y.to(torch.float32).contiguous()
        B, N = y.shape
        T = self.num_frames(N, key_shift, speed)
        eng = self._engine(y.device)
        out = torch.empty((B, self.n_mels, T), device=y.device, dtype=torch.float32)
        with eng.lock:
            _lib.check(_lib.lib().fdx_mel_forward(eng.h, _lib.ptr(y), B, N, float(key_shift), float(speed), int(log_mode),
                                                  _lib.ptr(out), _lib.stream_ptr(y.device)), eng.h)
        return out
