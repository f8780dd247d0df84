#!/usr/bin/env python3
"""RefineGAN generator alone: ms per 10 s utterance (hop 256 -> T = 1722), device RNG."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fish_diffusion_amd import RefineGANGenerator  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen = RefineGANGenerator()
gen.remove_weight_norm()
with torch.no_grad():
    for k, p in gen.named_parameters():
        if p.dim() == 3:
            p.copy_(torch.randn_like(p) * (1.0 / (p.shape[1] * p.shape[2])) ** 0.5)
        elif k.endswith("weight"):
            p.fill_(0.1)
gen = gen.to(dev).eval()
gen.rng = "philox"
T = int(10 * 44100) // 256
mel = torch.randn(B, 128, T, device=dev) * 0.5 - 2.0
f0 = bench.synth_inputs(B, T, dev, 1)[1]
for _ in range(3):
    gen(mel, f0)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    gen(mel, f0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
# 2*MAC of every conv (generator.py:333-423)
c, L = 16, T * 256
fl = 2 * 16 * 7 * L
length = L
for r in (2, 2, 8, 8):
    length //= r
    fl += 2 * length * 7 * (2 * c * c + 5 * (2 * c) ** 2)
    c *= 2
fl += 2 * T * 7 * 128 * c
c *= 2
fl += 2 * (T * 8) * c * 64
length = T
for r in (8, 8, 2, 2):
    length *= r
    n = c // 2
    fl += 2 * length * (7 * (c + c // 4) * n + sum(6 * k * n * n for k in (3, 7, 11)))
    c = n
fl += 2 * L * 7 * c
fl *= B
print(f"B={B}: {dt*1e3:.3f} ms per batch ({B * 10 / dt:.0f}x real-time), {fl/1e9:.1f} GFLOP -> {fl/dt/1e12:.1f} TFLOP/s ({fl/dt/1e12/157.3*100:.1f}% of fp32 peak)")
