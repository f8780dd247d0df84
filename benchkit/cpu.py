"""The `cpu_baseline` leg of bench.py: the pinned CPU oracle (oracle/: test infrastructure, imported HERE and nowhere else in the bench)
timed on the GPU box's host cores on a bounded sample of the same workload."""
from __future__ import annotations

import os
import time

import torch

from .flops import WN_CFG
from .workloads import synth_f0


def usable_cores() -> int:
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box is a
    256-thread EPYC with a 16-CPU quota: 256 torch threads there oversubscribe 16x and run ~5x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_denoiser(diff):
    """The oracle's restatement of whichever denoiser `diff` holds, on its CPU copy of the weights."""
    kind = getattr(diff.denoise_fn, "_KIND", "wavenet")
    sd = {k: v.detach().cpu() for k, v in diff.denoise_fn.state_dict().items()}
    if kind == "convnext":
        from oracle import convnext_ref
        n, cyc = diff.denoise_fn._desc.num_layers, diff.denoise_fn._desc.dilation_cycle
        return lambda x, t, c, xm, cm: convnext_ref.convnext_forward(sd, x, t, c, xm, cm, num_layers=n, dilation_cycle=cyc)
    if kind == "tfdec":
        from oracle import tfdec_ref
        n = diff.denoise_fn.n_layers
        return lambda x, t, c, xm, cm: tfdec_ref.tfdec_forward(sd, x, t, c, xm, cm, num_layers=n)
    from oracle import wavenet_ref
    return lambda x, t, c, xm, cm: wavenet_ref.wavenet_forward(sd, x, t, c, xm, cm, residual_layers=WN_CFG["residual_layers"],
                                                               dilation_cycle=WN_CFG["dilation_cycle"])


CPU_REPEATS = 3   # 1 warm-up + 3 timed runs, median


def cpu_chain(diff, voc, nsf, T, n_steps, sample_steps, predictor=None, repeats=CPU_REPEATS):
    """The oracle chain on this box's host cores for ONE utterance of T frames: `sample_steps` of the `n_steps` denoiser calls at
    full length (the rest extrapolated linearly: every step is the same call) + the full vocoder pass.  Protocol of BASELINE.md
    section 3: one warm-up pass (a short sampler run + one vocoder pass: thread pool, MKL-DNN primitive caches, page faults), then
    `repeats` timed passes; returns the MEDIAN pass (by total) and every pass's (denoise, vocoder) seconds."""
    from oracle import nsf_hifigan_ref, sampler_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    hop = nsf["hop_size"] if nsf else 512
    runs = []
    with torch.no_grad():
        den = cpu_denoiser(diff) if diff is not None else None
        feats, x0 = torch.randn(1, T, 256, generator=g), torch.randn(1, 128, T, generator=g)
        if voc is not None:
            gsd = {k: v.detach().cpu() for k, v in voc.model.state_dict().items()}
            f0 = synth_f0(T, nsf["sampling_rate"] / hop)[None]
            ri = torch.rand(1, 9, generator=g)
            sn = torch.randn(1, T * hop, 9, generator=g)
        for r in range(repeats + 1):
            warm = r == 0
            ss = min(sample_steps, 5) if warm else sample_steps
            t_den = 0.0
            if den is not None:
                kw = {}
                if predictor == "naive":
                    kw = dict(predictor="naive", step_noise=torch.randn(ss, 1, 128, T, generator=g))
                t0 = time.perf_counter()
                mel = sampler_ref.diffusion_sample(den, feats, x_init=x0, sampler_interval=1000 // ss, **kw)
                t_den = (time.perf_counter() - t0) / ss * n_steps
                melv = 2.30259 * mel.transpose(1, 2)
            else:
                melv = torch.randn(1, 128, T, generator=g) * 0.5 - 2.0
            t_voc = 0.0
            if voc is not None:
                t0 = time.perf_counter()
                nsf_hifigan_ref.generator_forward(gsd, nsf, melv, f0, ri, sn)
                t_voc = time.perf_counter() - t0
            if not warm:
                runs.append((t_den, t_voc))
    med = sorted(runs, key=lambda p: p[0] + p[1])[len(runs) // 2]
    return med[0], med[1], cores, runs


def refinegan_chain(model, cfg, T, repeats):
    """HiFiSinger front end + RefineGAN generator of ONE item on the host cores (oracle/features_ref.py, oracle/refinegan_ref.py)."""
    from oracle import features_ref, refinegan_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    hop = cfg["hop_length"]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    fsd = {k: v for k, v in sd.items() if not k.startswith("encoder.")}
    gsd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    contents = torch.randn(1, T, 768, generator=g)
    f0 = synth_f0(T, cfg["sampling_rate"] / hop)[None, None]
    noises = [torch.randn(shape, generator=g) for shape in refinegan_ref.noise_shapes(cfg, 1, T)]
    runs = []
    with torch.no_grad():
        for r in range(repeats + 1):
            t0 = time.perf_counter()
            feats = features_ref.hifisinger_features(fsd, contents, torch.zeros(1, dtype=torch.long), torch.tensor([T]), T, pitch_shift=torch.zeros(1, 1),
                                                     energy=torch.rand(1, T, 1, generator=g) * 0.1)
            mel = feats["features"] if isinstance(feats, dict) else feats
            refinegan_ref.generator_forward(gsd, cfg, mel.transpose(1, 2).contiguous(), f0, noises)
            if r:
                runs.append((0.0, time.perf_counter() - t0))
    med = sorted(runs, key=lambda p: p[1])[len(runs) // 2]
    return med[0], med[1], cores, runs


def cpu_baseline_leg(w, cfg, args, value, quick=False):
    """The oracle chain timed on this box's host cores on a bounded sample of the same workload (rank 0, one GPU).  `quick` (the sub-lines of
    the default run): 1 warm-up + ONE timed pass over a handful of sampler steps -- a few seconds per line, extrapolated like the full leg."""
    diff, voc, nsf, T, n_steps, B, hop = w.diff, w.voc, w.nsf, w.T, w.n_steps, w.B, w.hop
    seconds = w.seconds
    rep = 1 if quick else CPU_REPEATS
    if cfg == "headline":
        ss = args.cpu_sample_steps or (5 if quick else 100)
        td, tv, cores, runs = cpu_chain(diff, voc, nsf, T, n_steps, ss, repeats=rep)
        sample = (f"1 x {seconds:g} s utterance (T={T}): {ss} of {n_steps} UniPC steps timed ({td / n_steps * 1e3:.0f} ms/step"
                  + ("" if ss == n_steps else f", extrapolated x{n_steps / ss:g}") + f") + full NSF-HiFiGAN pass ({tv:.2f} s)")
        cpu_audio = seconds
    elif cfg == "vocoder":
        td, tv, cores, runs = cpu_chain(None, voc, nsf, T, 0, 0, repeats=rep)
        sample = f"1 of the {B} x {seconds:g} s mels (T={T}): one full NSF-HiFiGAN config_v1_256 pass ({tv:.2f} s)"
        cpu_audio = seconds
    elif cfg == "sharded":
        ss = args.cpu_sample_steps or (5 if quick else 20)
        Tm = sorted(w.lens[i] for i in w.mine)[len(w.mine) // 2]
        td, tv, cores, runs = cpu_chain(diff, voc, nsf, Tm, n_steps, ss, repeats=rep)
        sample = (f"1 utterance of median length (T={Tm}) run alone: {ss} of {n_steps} UniPC steps timed, extrapolated x{n_steps / ss:g}, + full "
                  f"NSF-HiFiGAN pass ({tv:.2f} s)")
        cpu_audio = Tm * hop / 44100.0
    elif cfg == "ddpm1000":
        ss = args.cpu_sample_steps or (5 if quick else 50)
        td, tv, cores, runs = cpu_chain(diff, voc, nsf, T, n_steps, ss, predictor="naive", repeats=rep)
        sample = (f"1 of the {B} x {seconds:g} s utterances (T={T}): {ss} of {n_steps} DDPM steps timed ({td / n_steps * 1e3:.0f} ms/step, "
                  f"extrapolated x{n_steps / ss:g}) + full NSF-HiFiGAN pass ({tv:.2f} s); fp32")
        cpu_audio = seconds
    elif cfg == "hifisinger_v2":
        from .flops import RG_HIFISINGER
        td, tv, cores, runs = refinegan_chain(w.keep, RG_HIFISINGER, T, rep)
        sample = f"1 of the {B} x {seconds:g} s items (T={T}): front end + one full RefineGAN generator pass ({tv:.2f} s)"
        cpu_audio = seconds
    elif cfg in ("convnext", "tfdec"):
        ss = args.cpu_sample_steps or (5 if quick else 20)
        td, tv, cores, runs = cpu_chain(diff, None, None, T, n_steps, ss, repeats=rep)
        sample = (f"1 x {seconds:g} s utterance (T={T}): {ss} of {n_steps} UniPC steps over the oracle's {cfg} restatement timed ({td / n_steps * 1e3:.0f} ms/step, "
                  f"extrapolated x{n_steps / ss:g}); mel only, like the line it stands beside")
        cpu_audio = seconds
    else:
        raise ValueError(cfg)
    cb = {"value": round(cpu_audio / (td + tv), 4), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
          "sample": (sample + f"; torch {torch.__version__} CPU, {cores} threads; 1 warm-up pass + "
                     + (f"median of {len(runs)} timed passes" if len(runs) > 1 else "1 timed pass")),
          "denoise_s": round(td, 4), "vocoder_s": round(tv, 4),
          "protocol": ("BASELINE.md section 3: 1 warm-up + 3 timed, median" if not quick else
                       "bounded: 1 warm-up + 1 timed pass (the full protocol runs under `--config <name>`)"),
          "note": REF_VS_PORT_NOTE,
          "runs_s": [[round(a, 4), round(b, 4)] for a, b in runs],
          "runs_value": [round(cpu_audio / (a + b), 4) for a, b in runs]}
    return cb, round(value / cb["value"], 1)


def _ref_vs_port_note():
    """`kind` is "port": the GPU box has no reference tree.  The committed build-box timing of the REAL reference classes beside this port backs the label."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_cpu_reference_vs_port.json")))
    if not files:
        return "the pinned oracle port (torch CPU ops); no reference-vs-port timing committed"
    try:
        d = json.load(open(files[-1]))
        return (f"the pinned oracle port (torch CPU ops, outputs torch.equal to the reference's).  {os.path.relpath(files[-1], root)}: the REAL reference classes "
                f"(GaussianDiffusion + WaveNet, nsf_hifigan Generator) beside this port on the build box, same inputs / weights / threads: port time / reference "
                f"time = {d['port_over_reference_time']} ({d['reference']['x_realtime']} vs {d['port']['x_realtime']} x real-time on {d['box']})")
    except Exception as e:   # noqa: BLE001
        return f"the pinned oracle port; {files[-1]} unreadable ({e})"


REF_VS_PORT_NOTE = _ref_vs_port_note()
