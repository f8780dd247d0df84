#!/usr/bin/env python3
"""Host-side cost of one step: how long the host takes to ENQUEUE the sampler / vocoder vs how long the GPU runs."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
diff, voc = bench.seeded_modules(dev)
voc.model.rng = "philox"
feats, f0 = bench.synth_inputs(1, 861, dev, 1)
for _ in range(2):
    bench.one_step(diff, voc, feats, f0, 10)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    mel = diff(feats, sampler_interval=10)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    wav = voc.model(mel.transpose(1, 2), f0, mel_scale=2.30259)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"sampler: host enqueue {1e3*(t1-t0):7.2f} ms, total {1e3*(t2-t0):7.2f} ms | vocoder: enqueue {1e3*(t3-t2):6.2f} ms, total {1e3*(t4-t2):6.2f} ms")
