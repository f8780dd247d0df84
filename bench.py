#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): audio-seconds/sec (100-step denoise + NSF-HiFiGAN, 44.1 kHz).
One "step" = one batch of synthetic utterances through the whole path with the inputs already in HBM:
    features [B,T,256] + f0 [B,T]  ->  x_T ~ N(0,1)  ->  100-step UniPC sampler driving the WaveNet denoiser
    ->  denorm  ->  NSF-HiFiGAN (config_v1, hop 512)  ->  waveform [B, T*512]
Workload at N=1 = BASELINE configs[1]: svc_hubert_soft arch (diff_svc_v2: C=512, 20 layers), batch=1, 10 s @ 44.1 kHz
(T=861), sampler_interval=10.  With N>1 every rank runs the same per-GPU workload on its own utterances
(weak scaling; utterances are independent -- SURVEY 8e); the only collective is the start-up RCCL broadcast of
the packed weights (outside the timed region) and the max-over-ranks of the wall time.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the dilated-conv + gate MFMA kernel of the
residual block, 60 % of all FLOPs): algorithmic FLOPs per launch / its average launch duration measured with HIP
events on the launch stream inside the timed region; the bound is the fp32 MFMA roof (157.3 TFLOP/s: the path is
fp32 for parity and compute-bound by 8-19x).  `cpu_baseline` is the CPU oracle (the pinned restatement of
the reference's PyTorch path, torch CPU ops) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WN_CFG = dict(mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=20, dilation_cycle=4,
              use_linear_bias=True)  # configs/_base_/archs/diff_svc_v2.py:27-35
NSF_V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 2, 2],
              upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
              resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=128, n_fft=2048, hop_size=512,
              win_size=2048, sampling_rate=44100, fmin=40, fmax=16000)  # tools/nsf_hifigan/config_v1.json
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 matrix == vector peak
PEAK_HBM_GBS = 8000.0


def wavenet_flops_per_frame(c=WN_CFG):
    C, L, M, E = c["residual_channels"], c["residual_layers"], c["mel_channels"], c["d_encoder"]
    return 2.0 * (M * C + L * (3 * C * 2 * C + E * 2 * C + C * 2 * C) + C * C + C * M)


def nsf_flops_per_sample(h=NSF_V1):
    """2*MAC of every conv in Generator.forward per OUTPUT sample (1.2737 MFLOP for config_v1)."""
    hop = h["hop_size"]
    C0 = h["upsample_initial_channel"]
    total = 2.0 * h["num_mels"] * C0 * 7 / hop
    rate = 1.0 / hop
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, cout = C0 >> i, C0 >> (i + 1)
        total += 2.0 * cin * cout * k * rate          # ConvTranspose1d: k taps per INPUT sample
        rate *= u
        s = int(round(1.0 / rate))                     # remaining upsampling = noise conv stride
        total += 2.0 * cout * (2 * s if s > 1 else 1) * rate
        for kk, dils in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
            n_convs = len(dils) * (2 if h["resblock"] == "1" else 1)
            total += 2.0 * cout * cout * kk * n_convs * rate
    total += 2.0 * cout * 7
    return total


def seeded_modules(device, seed=1234):
    """Random-init weights of the named architecture (no checkpoints exist offline).  The reference zero-inits the
    final projection (wavenet.py:192) and N(0,0.01)-inits the vocoder, which would make every activation ~0: use
    fan-in scaled draws so the data flowing through the kernels has O(1) magnitude (DVFS sees realistic toggling)."""
    from fish_diffusion_amd import DIFFUSIONS, NsfHifiGAN
    from fish_diffusion_amd.nsf_hifigan import generator_param_table
    torch.manual_seed(seed)
    diff = DIFFUSIONS.build(dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", **WN_CFG),
                                 spec_min=[-5], spec_max=[0], sampler_interval=10))
    torch.nn.init.normal_(diff.denoise_fn.output_projection.conv.weight, std=0.02)
    g = torch.Generator().manual_seed(seed + 1)
    state = {}
    for key, shape, _ in generator_param_table(NSF_V1):
        if key.endswith("bias") or len(shape) < 3:
            state[key] = torch.randn(shape, generator=g) * 0.01
        else:
            fan_in = shape[1] * shape[2] if "ups." not in key else shape[0] * shape[2] / max(1, NSF_V1["upsample_rates"][int(key.split(".")[1])])
            state[key] = torch.randn(shape, generator=g) * math.sqrt(1.0 / max(1.0, fan_in))
    voc = NsfHifiGAN.from_state(NSF_V1, state, use_natural_log=False)
    return diff.to(device).eval(), voc.to(device).eval()


def synth_inputs(B, T, device, seed):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, 256, generator=g)
    t = torch.arange(T, dtype=torch.float32) / (44100 / 512)
    f0 = 220.0 * torch.pow(2.0, 0.3 * torch.sin(2 * math.pi * 0.7 * t))
    f0[100:130] = 0.0
    return feats.to(device), f0[None].repeat(B, 1).contiguous().to(device)


def one_step(diff, voc, feats, f0, interval, streams=None):
    """One utterance batch: sampler, then vocoder.  With `streams` = (s_den, s_voc) the two stages are enqueued on
    separate HIP streams (vocoder waits on an event): consecutive steps are independent utterances, so the vocoder of
    step k overlaps the sampler of step k+1 and fills the CUs the batch-1 denoiser leaves idle (224 tiles / 256 CUs)."""
    if streams is None:
        mel = diff(feats, sampler_interval=interval)                       # [B, T, M] (log10-scale mel, diff_svc_v2)
        return voc.model(mel.transpose(1, 2), f0, mel_scale=2.30259)     # spec2wav for a batch (nsf_hifigan.py:72-85)
    s_den, s_voc = streams
    with torch.cuda.stream(s_den):
        mel = diff(feats, sampler_interval=interval)
        done = torch.cuda.Event()
        done.record(s_den)
    with torch.cuda.stream(s_voc):
        s_voc.wait_event(done)
        mel.record_stream(s_voc)
        wav = voc.model(mel.transpose(1, 2), f0, mel_scale=2.30259)
    return wav


def pmc_traffic(batch: int, T: int):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (tools/pmc_traffic.py ->
    profiles/*_pmc_traffic.json; FETCH_SIZE and WRITE_SIZE need separate passes, so bench.py cannot collect them
    itself).  Only valid for the configuration the counters were collected on (batch 1, T = 861)."""
    if batch != 1 or T != 861:
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    for k, v in d["kernels"].items():
        if "EpiGate" in k:
            return v["hbm_bytes"], os.path.relpath(files[-1], ROOT)
    return None, None


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


def cpu_baseline(diff, voc, T, seconds, n_steps, sample_steps):
    """The oracle (restatement of the reference PyTorch path, same torch CPU ops) on this box's host cores.
    Bounded sample: `sample_steps` of the `n_steps` UniPC steps at full length + the full vocoder pass; the rest of
    the sampler is extrapolated linearly (every step is the same denoiser call)."""
    from oracle import nsf_hifigan_ref, sampler_ref, wavenet_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in diff.denoise_fn.state_dict().items()}
    gsd = {k: v.detach().cpu() for k, v in voc.model.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    feats, x0 = torch.randn(1, T, 256, generator=g), torch.randn(1, 128, T, generator=g)
    den = lambda x, t, c, xm, cm: wavenet_ref.wavenet_forward(sd, x, t, c, xm, cm, residual_layers=WN_CFG["residual_layers"],  # noqa: E731
                                                              dilation_cycle=WN_CFG["dilation_cycle"])
    with torch.no_grad():
        den(x0, torch.tensor([500.0]), feats.transpose(1, 2), None, None)   # warm-up (thread pool, MKL-DNN primitives)
        t0 = time.perf_counter()
        mel = sampler_ref.diffusion_sample(den, feats, x_init=x0, sampler_interval=1000 // sample_steps)
        t_den = (time.perf_counter() - t0) / sample_steps * n_steps
        f0 = torch.full((1, T), 220.0)
        ri = torch.rand(1, 9, generator=g)
        sn = torch.randn(1, T * 512, 9, generator=g)
        t0 = time.perf_counter()
        nsf_hifigan_ref.generator_forward(gsd, NSF_V1, 2.30259 * mel.transpose(1, 2), f0, ri, sn)
        t_voc = time.perf_counter() - t0
    return {"value": seconds / (t_den + t_voc), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "sample": f"1 x {seconds:g} s utterance (T={T}): {sample_steps} of {n_steps} UniPC steps timed ({t_den / n_steps * 1e3:.0f} ms/step"
                      + ("" if sample_steps == n_steps else f", extrapolated x{n_steps / sample_steps:g}")
                      + f") + full NSF-HiFiGAN pass ({t_voc:.2f} s); torch {torch.__version__} CPU, {cores} threads",
            "denoise_s": t_den, "vocoder_s": t_voc}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step (configs[1]: 1)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--interval", type=int, default=10, help="sampler_interval: 10 => 100 UniPC steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-steps", type=int, default=100, help="UniPC steps the CPU baseline actually runs (100 = the whole "
                    "workload, ~10 s on 16 cores; fewer steps are extrapolated linearly)")
    ap.add_argument("--overlap", action="store_true", help="sampler and vocoder on separate HIP streams (vocoder of utterance k "
                    "overlaps the sampler of k+1).  Measured on MI355X: 96.3 vs 95.0 ms per step -- no gain, the co-running "
                    "vocoder kernels slow the denoiser's by as much as they hide; off by default.")
    ap.add_argument("--no-prof", action="store_true", help="do not time the dominant kernel with HIP events")
    ap.add_argument("--storage", choices=("fp32", "bf16"), default="fp32",
                    help="bf16: the WaveNet's OPT-IN bf16 storage mode (BASELINE configs[4]); not parity-grade, reported as its own dtype, "
                         "no roofline / cpu_baseline -- the contract's line is the fp32 default")
    ap.add_argument("--prof-stride", type=int, default=7, help="time every N-th launch of the dominant kernel (7 is co-prime "
                    "with the 20 layers, so every layer / dilation is sampled)")
    args = ap.parse_args()
    if args.storage == "bf16":
        args.no_prof = args.no_cpu_baseline = True

    from fish_diffusion_amd import _lib, dist as fdist
    import ctypes as C

    rank, local_rank, world = fdist.init_process_group()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    T = int(args.seconds * 44100) // 512
    n_steps = 1000 // args.interval
    diff, voc = seeded_modules(dev)
    # rank 0's packed weights reach the other ranks by one RCCL broadcast (outside the timed region)
    t0 = time.perf_counter()
    fdist.broadcast_model_weights(diff.denoise_fn, voc.model, dev, src=0)
    torch.cuda.synchronize()
    if args.storage == "bf16":   # after the (fp32) arenas are in place: every rank packs its own bf16 copy of the two GEMMs' weights
        diff.denoise_fn.storage = "bf16"
    t_bcast = time.perf_counter() - t0
    voc.model.rng = "philox"          # perf mode: source noise drawn on the device inside the library
    feats, f0 = synth_inputs(args.batch, T, dev, 1234 + rank)

    eng = diff.denoise_fn.engine(dev)

    def sync_barrier():
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev)) if args.overlap else None
    if streams:   # inputs were produced on the default stream
        for st_ in streams:
            st_.wait_stream(torch.cuda.current_stream(dev))
    for _ in range(args.warmup):
        one_step(diff, voc, feats, f0, args.interval, streams)
    sync_barrier()
    if not args.no_prof:
        _lib.check(_lib.lib().fdx_prof_enable(eng.h, args.prof_stride), eng.h)
        sync_barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        wav = one_step(diff, voc, feats, f0, args.interval, streams)
        if k == 0 and not args.no_prof:   # the dominant kernel is timed on the first timed step only (it needs the eager
            _lib.check(_lib.lib().fdx_prof_enable(eng.h, -1), eng.h)   # launch path); the other steps replay the hipGraph
    sync_barrier()
    dt = time.perf_counter() - t0
    dt = fdist.barrier_max(dt, dev)

    roofline = None
    if not args.no_prof:
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        _lib.check(_lib.lib().fdx_prof_read(eng.h, C.byref(n), C.byref(ms), C.byref(fl)), eng.h)
        _lib.check(_lib.lib().fdx_prof_enable(eng.h, 0), eng.h)
        if n.value:
            avg_ms = raw_ms = ms.value / n.value   # per-dispatch begin/end stamps (hipExtLaunchKernel events)
            ach = fl.value / (avg_ms * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic(args.batch, T)
            C_, M_ = WN_CFG["residual_channels"], args.batch * T
            alg_bytes = 4 * (2 * C_ * 3 * C_ + C_ * M_ + 2 * C_ * M_ + C_ * M_)   # weights + Y in + conditioner slab in + Z out
            kname = ("convgemm_kernel<2,splitK,EpiGate> (v_mfma_f32_32x32x2_f32)" if os.environ.get("FDX_RESBLOCK_MFMA") == "32"
                     else "convgemm16_kernel<EpiGate16> (v_mfma_f32_16x16x4_f32)")
            roofline = {"bound": "mfma", "kernel": kname + ": dilated conv k=3 + gate of the residual block",
                        "achieved": round(ach, 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_TFLOPS, 4),
                        "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)",
                        "traffic_source": traffic_src, "algorithmic_bytes": alg_bytes,
                        "hbm_fraction": (round(traffic / (avg_ms * 1e-3) / 8.0e12, 4) if traffic else None), "launches_timed": n.value, "sampling": f"every {args.prof_stride}th launch of the first timed step", "avg_launch_us": round(avg_ms * 1e3, 2),
                        "timing": "hipExtLaunchKernel start/stop events on the launch stream, timed region",
                        "flops_per_launch": fl.value}

    # per-stage split (outside the timed region)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    mel = diff(feats, sampler_interval=args.interval)
    ev[1].record()
    voc.model(mel.transpose(1, 2), f0, mel_scale=2.30259)
    ev[2].record()
    torch.cuda.synchronize()
    den_ms, voc_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])

    audio_s = args.batch * T * 512 / 44100.0
    value = world * args.steps * audio_s / dt
    flops_step = args.batch * (wavenet_flops_per_frame() * T * n_steps + nsf_flops_per_sample() * T * 512)
    e2e_tflops = flops_step * args.steps / dt / 1e12

    out = {
        "metric": "audio-seconds/sec (100-step denoise + NSF-HiFiGAN, 44.1 kHz)" if n_steps == 100 else
                  f"audio-seconds/sec ({n_steps}-step denoise + NSF-HiFiGAN, 44.1 kHz)",
        "value": round(value, 3), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.storage == "fp32" else "bf16 storage / f32 accumulate (opt-in mode, not parity-grade)", "data": "synthetic (seeded N(0,1) features, vibrato f0 with an unvoiced gap; random-init weights of the named architecture)",
        "config": {"workload": f"svc_hubert_soft (diff_svc_v2 WaveNet C=512 x 20 layers) {n_steps}-step UniPC + NSF-HiFiGAN config_v1 (hop 512), "
                               f"batch={args.batch} x {args.seconds:g} s @44.1 kHz (T={T}) per GPU",
                   "batch_per_gpu": args.batch, "frames": T, "sampler": "unipc", "sampler_steps": n_steps,
                   "parallelism": f"utterance-sharded x{world} (no per-step collective)",
                   "streams": "sampler and vocoder on separate HIP streams (vocoder of utterance k overlaps sampler of k+1)"
                              if streams else "single stream"},
        "per_gpu": round(value / world, 3), "x_realtime_per_gpu": round(value / world, 3),
        "stages_ms": {"denoise": round(den_ms, 2), "vocoder": round(voc_ms, 2)},
        "end_to_end": {"tflops": round(e2e_tflops, 3), "frac_of_f32_peak": round(e2e_tflops / PEAK_F32_TFLOPS, 4),
                       "algorithmic_flops_per_step": flops_step},
        "weights_bcast_s": round(t_bcast, 4),
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(diff, voc, T, args.seconds, n_steps, args.cpu_sample_steps)
        out["cpu_baseline"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cb.items()}
        out["gpu_over_cpu"] = round(value / cb["value"], 1)
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
