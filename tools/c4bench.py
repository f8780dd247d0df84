#!/usr/bin/env python3
"""BASELINE configs[3] on the GPUs at hand: 64 utterances of 6-10 s (T in [516, 861]), sharded over `--world` ranks
(this process plays rank `--rank`), micro-batches of <= 8 with masks, 100-step UniPC + NSF-HiFiGAN.  Prints this
rank's audio-seconds per second (per-GPU throughput of the sharded job)."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fish_diffusion_amd import pipeline  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--max-batch", type=int, default=8)
ap.add_argument("--n", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda", 0)
diff, voc = bench.seeded_modules(dev)
voc.model.rng = "philox"
g = torch.Generator().manual_seed(4)
lens = torch.randint(516, 862, (a.n,), generator=g).tolist()
feats = [torch.randn(n, 256, generator=g).to(dev) for n in lens]
f0s = [bench.synth_inputs(1, n, dev, 0)[1][0] for n in lens]
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = pipeline.synthesize(diff, voc, feats, f0s, max_batch=a.max_batch, sampler_interval=10, rank=a.rank, world=a.world)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    audio = sum(w.numel() for _, _, w in res) / 44100.0
    print(f"pass {rep}: rank {a.rank}/{a.world}: {len(res)} utterances, {audio:.1f} s of audio in {dt*1e3:.1f} ms -> {audio/dt:.1f}x real-time per GPU "
          f"(batches: {[len(b) for b in pipeline.make_batches([lens[i] for i in pipeline.fdist.shard_utterances(lens, a.rank, a.world)], a.max_batch)]})")
