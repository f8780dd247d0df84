#!/usr/bin/env python3
"""TransformerDecoderDenoiser (dim 512, 12 layers) under the UniPC sampler: ms per 10 s utterance batch, TFLOP/s."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fish_diffusion_amd import GaussianDiffusion  # noqa: E402
from oracle import tfdec_ref  # noqa: E402  (seeded weights only)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
interval = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
cfg = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=12)
diff = GaussianDiffusion(dict(type="TransformerDecoderDenoiser", **cfg), spec_min=[-5], spec_max=[0])
diff.denoise_fn.load_state_dict(tfdec_ref.seeded_state(1, **cfg))
diff = diff.to(dev).eval()
T = 861
feats = torch.randn(B, T, 256, device=dev)
x0 = torch.randn(B, 128, T, device=dev)
for _ in range(2):
    diff(feats, sampler_interval=interval, x_init=x0)
torch.cuda.synchronize()
N = 3
t0 = time.perf_counter()
for _ in range(N):
    diff(feats, sampler_interval=interval, x_init=x0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
D, H, L, M = 512, 2048, 12, 128
calls = 1000 // interval + 1
gemm = 2 * (M * H + H * D + L * (3 * D * D + D * D + D * D + 2 * D * D + D * D + 2 * D * H) + D * D + D * M)   # per frame
attn = L * 2 * 2 * 2 * T * D                                                                                      # per frame: QK^T and PV, 2 attentions
fl = calls * B * T * (gemm + attn)
print(f"B={B}: {dt*1e3:.2f} ms per batch ({calls} denoiser calls, {dt/calls*1e6:.1f} us each), {fl/dt/1e12:.1f} TFLOP/s "
      f"({fl/dt/1e12/157.3*100:.1f}% of fp32 peak)")
