#!/usr/bin/env python3
"""A SHORT run of one bench geometry for the rocprofv3 PMC passes that die on the full bench command (round 5: `--pmc FETCH_SIZE|WRITE_SIZE`
over `bench.py --config tfdec|sharded` segfaults inside rocprofv3; round 4: hung on 8 k incomplete dispatches).  Same kernels, same shapes, a
few hundred dispatches instead of 45 k: a 4-step UniPC run, eager (no hipGraph), so per-launch traffic of every kernel is what the passes see.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o pmc -- python tools/pmc_mini.py tfdec      (and WRITE_SIZE in a second pass)
    python tools/pmc_traffic.py <fetch_db> <write_db> tfdec > profiles/r05_tfdec_pmc_traffic.json

  tfdec    TransformerDecoderDenoiser (dim 512 x 12 layers), batch 1 x 861 frames               = bench.py --config tfdec
  convnext ConvNextDenoiser (dim 512 x 20 blocks), batch 1 x 861 frames                          = bench.py --config convnext  (round 6: the PMC
           passes over the full command die as well since its graph grew past ~10 k dispatches per run)
  sharded  WaveNet (C = 512 x 20 layers), one exact-ragged micro-batch of 8 utterances, longest 861 frames  = a micro-batch of --config sharded
"""
import os
import sys

os.environ["FDX_NO_GRAPH"] = "1"
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fish_diffusion_amd import GaussianDiffusion  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "tfdec"
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(4)
if cfg in ("tfdec", "convnext"):
    from benchkit.flops import CN_CFG, TD_CFG
    from benchkit.workloads import seeded_denoiser      # the bench's own random-init modules (no oracle import)
    diff = (seeded_denoiser("TransformerDecoderDenoiser", TD_CFG) if cfg == "tfdec" else seeded_denoiser("ConvNextDenoiser", CN_CFG)).to(dev).eval()
    feats = torch.randn(1, 861, 256, generator=g).to(dev)
    run = lambda: diff(feats, sampler_interval=250)   # noqa: E731
elif cfg == "sharded":
    from oracle import wavenet_ref
    c = dict(mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=20, dilation_cycle=4, use_linear_bias=True)
    diff = GaussianDiffusion(dict(type="WaveNetDenoiser", **c), spec_min=[-5], spec_max=[0])
    diff.denoise_fn.load_state_dict(wavenet_ref.seeded_wavenet_state(1234, **{k: v for k, v in c.items() if k != "dilation_cycle"}))
    diff = diff.to(dev).eval()
    # exactly rank 0's share of `bench.py --config sharded` (64 seeded lengths dealt longest-first over 8 ranks: ONE exact-ragged micro-batch of 8)
    from fish_diffusion_amd import dist as fdist
    all_lens = torch.randint(516, 862, (64,), generator=torch.Generator().manual_seed(4)).tolist()
    lens = [all_lens[i] for i in fdist.shard_utterances(all_lens, 0, 8)]
    feats = torch.randn(len(lens), max(lens), 256, generator=g).to(dev)
    run = lambda: diff(feats, sampler_interval=250, lengths=lens)   # noqa: E731
else:
    raise SystemExit(f"unknown config {cfg}")
for _ in range(3):
    run()
torch.cuda.synchronize()
print("done", cfg)
