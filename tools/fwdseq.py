#!/usr/bin/env python3
"""One WaveNet denoiser call (batch 1 x 10 s, full-size net) after warm-up: run under `rocprofv3 --kernel-trace` and dump the launch
sequence with `tools/prof_summary.py <db> --sequence 50` to see every launch of a sampler step in order."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
dev = torch.device("cuda", 0)
diff, _ = bench.seeded_modules(dev)
if len(sys.argv) > 1:
    diff.denoise_fn.storage = sys.argv[1]
net = diff.denoise_fn
T = 861
x, cond, t = torch.randn(1, 128, T, device=dev), torch.randn(1, 256, T, device=dev), torch.tensor([500.0], device=dev)
for _ in range(5):
    net(x, t, cond)
torch.cuda.synchronize()
net(x, t, cond)
torch.cuda.synchronize()
