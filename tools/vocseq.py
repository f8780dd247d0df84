#!/usr/bin/env python3
"""One NSF-HiFiGAN pass (batch B x 10 s) timed stage by stage with torch events is not possible from outside the library; this prints the
whole-pass time for hop 512 and hop 256 with the fused small-channel ResBlock kernel on (default) or off (FDX_NSF_FUSED=0)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
dev = torch.device("cuda", 0)
for name, nsf in (("config_v1 (hop 512)", bench.NSF_V1), ("config_v1_256 (hop 256)", bench.NSF_V1_256)):
    _, voc = bench.seeded_modules(dev, nsf=nsf, denoiser=False)
    voc.model.rng = "philox"
    hop = nsf["hop_size"]
    T = int(10 * 44100) // hop
    for B in (1, 8, 32):
        mel = torch.randn(B, 128, T, device=dev) * 0.5 - 2.0
        f0 = bench.synth_f0(T, 44100 / hop)[None].repeat(B, 1).contiguous().to(dev)
        for _ in range(3):
            voc.model(mel, f0)
        torch.cuda.synchronize()
        n = 10 if B < 32 else 4
        t0 = time.perf_counter()
        for _ in range(n):
            voc.model(mel, f0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        fl = bench.nsf_flops_per_sample(nsf) * T * hop * B
        print(f"FDX_NSF_FUSED={os.environ.get('FDX_NSF_FUSED', '1')} {name} B={B}: {dt * 1e3:.3f} ms per batch, {fl / dt / 1e12:.1f} TFLOP/s "
              f"({fl / dt / 1e12 / 157.3 * 100:.1f} % of fp32 peak), {B * 10 / dt:.0f}x real-time", flush=True)
