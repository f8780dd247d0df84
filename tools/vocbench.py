#!/usr/bin/env python3
"""NSF-HiFiGAN alone: ms per 10 s utterance (config_v1 hop 512, or --hop256), batch B."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
HOP256 = "--hop256" in sys.argv        # tools/nsf_hifigan/config_v1_256.json, what configs/vocoder_nsf_hifigan.py points at
dev = torch.device("cuda", 0)
NSF = bench.NSF_V1_256 if HOP256 else bench.NSF_V1
diff, voc = bench.seeded_modules(dev, nsf=NSF, denoiser=False)
voc.model.rng = "philox"
hop = NSF["hop_size"]
T = int(10 * 44100) // hop
mel = (torch.randn(B, 128, T, device=dev) * 0.5 - 2.0)
_, f0 = bench.synth_inputs(B, T, dev, 1)
f0 = f0.contiguous()
for _ in range(3):
    voc.model(mel, f0)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 10
for _ in range(N):
    voc.model(mel, f0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
fl = bench.nsf_flops_per_sample(NSF) * T * hop * B
print(f"B={B}: {dt*1e3:.3f} ms per batch, {fl/dt/1e12:.1f} TFLOP/s ({fl/dt/1e12/157.3*100:.1f}% of fp32 peak), {B*10/dt:.0f}x real-time")
