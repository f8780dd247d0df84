#!/usr/bin/env python3
"""The opt-in bf16 storage mode of the WaveNet denoiser next to fp32: ms per batch under the UniPC / DDPM sampler, and the
mel difference between the two on the same inputs (what the mode costs in accuracy over a whole schedule)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
interval = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pred = sys.argv[3] if len(sys.argv) > 3 else "unipc"
dev = torch.device("cuda", 0)
diff, _ = bench.seeded_modules(dev)
diff.step_rng = "philox"
T = 861
feats = bench.synth_inputs(B, T, dev, 0)[0]
x0 = torch.randn(B, 128, T, device=dev)
sn = torch.randn(1000 // interval, B, 128, T, device=dev) if pred == "naive" and B * (1000 // interval) <= 1000 else None
out = {}
for mode in ("fp32", "bf16"):
    diff.denoise_fn.storage = mode
    for _ in range(2):
        mel = diff(feats, sampler_interval=interval, noise_predictor=pred, x_init=x0, step_noise=sn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 3 if interval >= 10 else 1
    for _ in range(N):
        mel = diff(feats, sampler_interval=interval, noise_predictor=pred, x_init=x0, step_noise=sn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    out[mode] = (dt, mel)
    calls = 1000 // interval + (1 if pred != "naive" else 0)
    fl = calls * B * T * 95.18e6
    print(f"{mode}: B={B} {pred} {1000 // interval} steps: {dt*1e3:.1f} ms per batch ({dt/calls*1e6:.0f} us per call), {fl/dt/1e12:.1f} algorithmic TFLOP/s, "
          f"{B*10/dt:.0f}x real-time (denoiser only)")
a, b = out["fp32"][1], out["bf16"][1]
if sn is not None or pred != "naive":
    print(f"mel, bf16 storage vs fp32: max |diff| / max |mel| = {float((a - b).abs().max() / a.abs().max()):.3e}; "
          f"rms diff / rms mel = {float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.3e}")
print(f"speed-up {out['fp32'][0] / out['bf16'][0]:.2f}x")
