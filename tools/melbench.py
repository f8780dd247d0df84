#!/usr/bin/env python3
"""STFT/mel front end alone: ms per 10 s of audio (wav2spec, natural-log mel), batch B."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fish_diffusion_amd import PitchAdjustableMelSpectrogram, _lib  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
pam = PitchAdjustableMelSpectrogram()
wav = (torch.rand(B, 441000, device=dev) - 0.5)
for _ in range(3):
    pam(wav, log_mode=_lib.MEL_LN)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    m = pam(wav, log_mode=_lib.MEL_LN)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"B={B}: {dt*1e3:.3f} ms per batch of 10 s clips -> mel {tuple(m.shape)} ({B*10/dt:.0f}x real-time)")
