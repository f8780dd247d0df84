#!/usr/bin/env python3
"""Crossover of the three kernel families the fp32-class denoiser can run a launch on -- fp32 MFMA (convgemm16s), fp16-split on 64 x 64
tiles (f16s64) and on 128-wide tiles (bf16lds F16S) -- by (batch, frames): ms per 50 UniPC denoiser calls for each geometry on the
command line, in the mode the environment forces.  Run once per mode:

    python tools/f16s_cross.py fp32                      1x215 1x430 1x861 2x861 ...
    FDX_F16S_SMALL=2 FDX_BF16_LDS=1000000000 python tools/f16s_cross.py fp16x3 ...     (small tiles for every geometry)
    FDX_BF16_LDS=1 python tools/f16s_cross.py fp16x3 ...                                (128-wide tiles for every geometry)
"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

mode = sys.argv[1]
dev = torch.device("cuda", 0)
diff, _ = bench.seeded_modules(dev)
diff.denoise_fn.storage = mode
tag = f"{mode} small={os.environ.get('FDX_F16S_SMALL', '-')} lds={os.environ.get('FDX_BF16_LDS', '-')} nst={os.environ.get('FDX_F16S_NST', '-')}/{os.environ.get('FDX_F16S_NST_O', '-')}"
for geo in sys.argv[2:]:
    B, T = (int(v) for v in geo.split("x"))
    feats = bench.synth_inputs(B, T, dev, 0)[0]
    x0 = torch.randn(B, 128, T, device=dev)
    for _ in range(2):
        diff(feats, sampler_interval=20, x_init=x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 4
    for _ in range(N):
        diff(feats, sampler_interval=20, x_init=x0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"CROSS {tag:44s} B={B:2d} T={T:4d}: {dt * 1e3:7.2f} ms per 50 steps  ({B * T * 512 / 44100 / (dt * 2):7.1f}x real-time at 100 steps, denoiser only)", flush=True)
