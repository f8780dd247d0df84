#!/usr/bin/env python3
"""BUILD-BOX ONLY (needs /root/reference): the REAL reference modules timed beside the oracle port that bench.py's `cpu_baseline`
runs on the GPU box (which has no reference tree), same inputs, same thread count -- so that "port ~ reference" is shown once.

    python tools/cpu_ref_vs_port.py > profiles/r03_cpu_reference_vs_port.json

Protocol of BASELINE.md section 3: 1 warm-up + 3 timed passes, median; configs[1] shape (1 x 10 s, T = 861, 100-step UniPC + NSF-HiFiGAN
config_v1); the denoiser leg runs `--steps` of the 100 sampler steps (default 20, extrapolated x5: every step is the same call)."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_import, nsf_hifigan_ref, sampler_ref, wavenet_ref  # noqa: E402
from oracle.make_golden import WN_FULL, build_ref_diffusion, oracle_denoiser, synth_f0  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--frames", type=int, default=861)
args = ap.parse_args()
R = _ref_import.load()
cores = os.cpu_count()
torch.set_num_threads(cores)
T, ss = args.frames, args.steps
sd = wavenet_ref.seeded_wavenet_state(1234, **{k: v for k, v in WN_FULL.items() if k != "dilation_cycle"})
diff = build_ref_diffusion(R, WN_FULL, sd)
den = oracle_denoiser(sd, WN_FULL)
h = nsf_hifigan_ref.CONFIG_V1
gsd = nsf_hifigan_ref.seeded_generator_state(55, h)
gen = R["Generator"](R["AttrDict"](h))
gen.remove_weight_norm()
gen.eval()
gen.load_state_dict(gsd, strict=True)
g = torch.Generator().manual_seed(0)
feats, x0 = torch.randn(1, T, 256, generator=g), torch.randn(1, 128, T, generator=g)
f0 = synth_f0(T)[None]
ri = torch.rand(1, 9, generator=g)
ri[:, 0] = 0
sn = torch.randn(1, T * 512, 9, generator=g)
runs = {"reference": [], "port": []}
with torch.no_grad():
    for r in range(4):
        n = 5 if r == 0 else ss
        torch.manual_seed(1)
        t0 = time.perf_counter()
        mel_r = diff(feats, sampler_interval=1000 // n)
        t1 = time.perf_counter()
        torch.manual_seed(2)
        gen(2.30259 * mel_r.transpose(1, 2), f0)
        t2 = time.perf_counter()
        torch.manual_seed(1)
        x_init = torch.randn(1, 128, T)
        t3 = time.perf_counter()
        mel_p = sampler_ref.diffusion_sample(den, feats, x_init=x_init, sampler_interval=1000 // n)
        t4 = time.perf_counter()
        nsf_hifigan_ref.generator_forward(gsd, h, 2.30259 * mel_p.transpose(1, 2), f0, ri, sn)
        t5 = time.perf_counter()
        assert torch.equal(mel_r, mel_p)
        if r:
            runs["reference"].append(((t1 - t0) / n * 100, t2 - t1))
            runs["port"].append(((t4 - t3) / n * 100, t5 - t4))
out = {"what": "REAL reference modules (GaussianDiffusion + WaveNet, nsf_hifigan Generator from /root/reference) vs the oracle port bench.py times as "
               "cpu_baseline, same inputs / weights / threads; mel outputs torch.equal",
       "box": f"build container, {cores} threads, torch {torch.__version__} CPU", "frames": T, "sampler_steps_timed": ss,
       "protocol": "1 warm-up + 3 timed passes, median by total; denoise seconds extrapolated to 100 steps"}
for k, v in runs.items():
    med = sorted(v, key=lambda p: p[0] + p[1])[1]
    out[k] = {"denoise_s_per_100_steps": round(med[0], 3), "vocoder_s": round(med[1], 3), "x_realtime": round(T * 512 / 44100 / (med[0] + med[1]), 4),
              "runs": [[round(a, 3), round(b, 3)] for a, b in v]}
out["port_over_reference_time"] = round((out["port"]["denoise_s_per_100_steps"] + out["port"]["vocoder_s"]) /
                                        (out["reference"]["denoise_s_per_100_steps"] + out["reference"]["vocoder_s"]), 4)
print(json.dumps(out, indent=1))
