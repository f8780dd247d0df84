#!/usr/bin/env python3
"""Where does a conv/GEMM launch spend its time?  Runs one WaveNet forward (T=861, full net) on the instrumented
library (`python -m fish_diffusion_amd._build --trace`, FDX_LIB_PATH=.../libfishdx_trace.so) and prints, per launch,
the per-wave shader-clock intervals:  start->loads issued, K loop, LDS write, barrier wait, epilogue."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FDX_LIB_PATH", os.path.join(ROOT, "fish_diffusion_amd", "csrc", "libfishdx_trace.so"))
from fish_diffusion_amd import DENOISERS, _lib  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
LAYERS = int(sys.argv[2]) if len(sys.argv) > 2 else 20      # fewer layers: do the weights (11 MB per layer) stay in the Infinity Cache?
STORAGE = sys.argv[3] if len(sys.argv) > 3 else "fp32"
T = 861
net = DENOISERS.build(dict(type="WaveNetDenoiser", mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=LAYERS,
                           dilation_cycle=4, use_linear_bias=True)).to(dev)
net.storage = STORAGE
x, cond, t = torch.randn(B, 128, T, device=dev), torch.randn(B, 256, T, device=dev), torch.tensor([500.0], device=dev)
for _ in range(6):
    net(x, t, cond)
torch.cuda.synchronize()
CAP, NL = 256 * max(B, 4), 12 if B > 4 else 64
buf = torch.zeros(NL * CAP * 32, dtype=torch.int64, device=dev)
lib = _lib.lib()
lib.fdx_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int]
# "dense" (4th argument): the traced forward is enqueued behind three untraced ones with no sync in between, so its launches are already queued
# when the GPU gets to them and run back to back (the hipGraph regime); default: one forward from an idle GPU (every launch waits for the host)
DENSE = len(sys.argv) > 4 and sys.argv[4] == "dense"
if DENSE:
    for _ in range(3):
        net(x, t, cond)
lib.fdx_debug_trace(C.c_void_p(buf.data_ptr()), NL, CAP)
net(x, t, cond)
torch.cuda.synchronize()
lib.fdx_debug_trace(None, 0, 0)
tr = buf.cpu().numpy().reshape(NL, CAP * 4, 8)  # 4 waves per workgroup
names = ["setup+prefetch", "K loop", "LDS write", "barrier wait", "epilogue"]
print(f"{'launch':>6} {'waves':>6} {'span':>8} | " + " ".join(f"{n:>14}" for n in names) + " | total/wave  first 4 slots (shader cycles, mean over waves; span = last end - first start; first 4 slots = t6 - t1: fill + 4 slots of MFMA issue)")
for l in range(NL):
    w = tr[l].reshape(-1, 8)
    w = w[w[:, 0] != 0].copy()
    if not len(w):
        continue
    for k in range(1, 6):        # kernels that do not stamp a phase (f16s64: no LDS-write / barrier phases): that phase reads 0
        z = w[:, k] == 0
        w[z, k] = w[z, k - 1]
    d = np.diff(w[:, :6].astype(np.int64), axis=1)
    span = int(w[:, 5].max() - w[:, 0].min())
    first = (w[:, 6].astype(np.int64) - w[:, 1].astype(np.int64))
    first = first[w[:, 6] != 0]
    rt = w[:, 7].astype(np.int64)
    rt = rt[(rt > 0) & (rt < 10 ** 7)]
    clk = f"  {d.sum(1).mean() / (rt.mean() * 10e-9) / 1e9:5.2f} GHz ({rt.mean() * 10:.0f} ns per wave)" if len(rt) else ""
    print(f"{l:>6} {len(w):>6} {span:>8} | " + " ".join(f"{d[:, i].mean():>14.0f}" for i in range(5)) + f" | {d.sum(1).mean():>10.0f} {first.mean() if len(first) else 0:>10.0f}{clk}")

# board power and sclk while the same forward runs back to back for ~1 s (which component of the real kernels draws the extra
# 200 W / reads 2.1 GHz?) -- the sampler bench.py uses
from benchkit.timing import SclkSampler  # noqa: E402
lib.fdx_debug_trace(None, 0, 0)
smp = SclkSampler(0)
with smp:
    for _ in range(400):
        net(x, t, cond)
    torch.cuda.synchronize()
rep = smp.report()
print("sclk / board power over 400 back-to-back forwards:", rep.get("mean"), "MHz,", (rep.get("board_power_w") or {}).get("mean"), "W")
