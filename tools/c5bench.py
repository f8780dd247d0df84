#!/usr/bin/env python3
"""BASELINE configs[4] as SURVEY F4 reads it, in the precision this library has (fp32): DDPM ("naive") sampler with
sampler_interval=1 => 1000 denoiser calls, multi-speaker front end (speaker-embedding table), one rank's share of the
batch-128 job over 8 GPUs = 16 utterances of 10 s, then the vocoder.  Third argument "bf16" selects the opt-in bf16 storage mode the config names (fp32 is the default)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fish_diffusion_amd import DiffSinger, pitch_to_scale  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
storage = sys.argv[3] if len(sys.argv) > 3 else "fp32"      # "bf16": the opt-in storage mode the config names
dev = torch.device("cuda", 0)
diff, voc = bench.seeded_modules(dev)
voc.model.rng = "philox"
cfg = dict(text_encoder=dict(type="NaiveProjectionEncoder", input_size=256, output_size=256),
           speaker_encoder=dict(type="NaiveProjectionEncoder", input_size=128, output_size=256, use_embedding=True),
           pitch_encoder=dict(type="NaiveProjectionEncoder", input_size=1, output_size=256, preprocessing=pitch_to_scale),
           diffusion=dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", **bench.WN_CFG), spec_min=[-5], spec_max=[0]))
m = DiffSinger(cfg).to(dev).eval()
m.diffusion = diff                                   # the seeded full-size denoiser of bench.py
diff.step_rng = "philox"                             # per-step noise from the device generator (no [1000, B, M, T] tensor)
diff.denoise_fn.storage = storage
T = 861
g = torch.Generator().manual_seed(5)
contents = torch.randn(B, T, 256, generator=g).to(dev)
f0 = bench.synth_inputs(B, T, dev, 0)[1]
spk = torch.randint(0, 128, (B,), generator=g).to(dev)
interval = 1000 // steps
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mel = m.infer(spk, contents, f0, sampler_interval=interval, noise_predictor="naive")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    wav = voc.model(mel.transpose(1, 2).contiguous(), f0, mel_scale=2.30259)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fl = steps * B * T * 95.18e6
    print(f"pass {rep} [{storage}]: B={B}, {steps} naive steps: denoise {1e3*(t1-t0):.0f} ms ({fl/(t1-t0)/1e12:.1f} TFLOP/s = {fl/(t1-t0)/157.3e12*100:.1f} % of the fp32 roof), "
          f"vocoder {1e3*(t2-t1):.0f} ms -> {B*10/(t2-t0):.2f}x real-time per GPU; finite: {bool(torch.isfinite(wav).all())}")
