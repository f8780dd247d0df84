#!/usr/bin/env python3
"""bench.py -- the hot path's metrics on MI355X, one JSON line per run.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config headline|vocoder|sharded|ddpm1000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--config` selects a BASELINE.json `configs[]` entry (default: the one the headline metric is quoted on):

  headline  (configs[1]; aliases 1, c1)   svc_hubert_soft arch (diff_svc_v2 WaveNet C=512 x 20 layers), batch 1 x 10 s @ 44.1 kHz
            (T=861), 100-step UniPC, then NSF-HiFiGAN config_v1 (hop 512).  One step = one utterance: features [1,T,256] + f0 ->
            x_T ~ N(0,1) -> sampler -> denorm -> vocoder -> waveform [1, T*512].  Every timed step gets a FRESH feature tensor,
            so the hoisted conditioner GEMM (`fdx_wavenet_prepare`) runs inside the timed region, as it does per utterance when
            serving.
  vocoder   (configs[2]; aliases 2, c2)   NSF-HiFiGAN only, tools/nsf_hifigan/config_v1_256.json (hop 256: what
            configs/vocoder_nsf_hifigan.py:31 points at), batch 32 x 10 s mel (T=1722).  One step = one batch.
  sharded   (configs[3]; aliases 3, c3, c4)   svc_content_vec (same model): 64 ragged utterances of 6-10 s, sharded longest-first
            over the ranks, masked micro-batches of <= 8 through `pipeline.synthesize`, 100-step UniPC + vocoder.  One step =
            this rank's whole shard.  With one process the shard is rank 0's share of an 8-way job (`--virtual-world`).
  ddpm1000  (configs[4]; aliases 4, c5)   1000-step DDPM ("naive", sampler_interval=1), speaker-embedding front end, 16 x 10 s
            utterances per GPU (= batch 128 over 8 GPUs), then the vocoder; fp32, or `--storage bf16` for the opt-in bf16 storage
            mode the config names (labelled as such, never parity-grade).

Inputs are resident in HBM when the timed region starts; `value` is the whole-job aggregate over all ranks (weak scaling: every
rank runs the same per-GPU workload on its own utterances; the only collective is the start-up RCCL broadcast of the packed
weights, outside the timed region, and the MAX over ranks of the wall time).  `pcie_inclusive` (headline only) repeats the step
with features / f0 starting in pinned host memory and the waveform copied back: reported beside `value`, never as `value`.

`roofline` is for the config's dominant kernel: algorithmic FLOPs per launch / average launch duration from HIP events recorded
on the launch stream inside the timed region (`fdx_prof_*`), against the fp32 MFMA roof (157.3 TFLOP/s: the path is fp32 for
parity and compute-bound); `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes of the same
command (profiles/).  `cpu_baseline` = the CPU oracle (pinned restatement of the reference's PyTorch path, torch CPU ops) timed
on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
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
NSF_V1_256 = dict(NSF_V1, upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], hop_size=256)  # config_v1_256.json
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 matrix == vector peak
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0
BASELINE_METRIC = "audio-seconds/sec/GPU (100-step denoise + NSF-HiFiGAN, 44.1 kHz)"   # BASELINE.json "metric", verbatim

ALIASES = {"headline": "headline", "1": "headline", "c1": "headline", "configs1": "headline",
           "vocoder": "vocoder", "2": "vocoder", "c2": "vocoder", "configs2": "vocoder",
           "sharded": "sharded", "3": "sharded", "c3": "sharded", "c4": "sharded", "configs3": "sharded",
           "ddpm1000": "ddpm1000", "4": "ddpm1000", "c5": "ddpm1000", "configs4": "ddpm1000"}
DEFAULT_STEPS = {"headline": (5, 2), "vocoder": (5, 2), "sharded": (3, 1), "ddpm1000": (2, 1)}


# ====================================================================================================== algorithmic work
def wavenet_flops_per_frame(c=WN_CFG):
    C_, L, M, E = c["residual_channels"], c["residual_layers"], c["mel_channels"], c["d_encoder"]
    return 2.0 * (M * C_ + L * (3 * C_ * 2 * C_ + E * 2 * C_ + C_ * 2 * C_) + C_ * C_ + C_ * M)


def wavenet_hoisted_flops_per_frame(c=WN_CFG):
    """The step-invariant part of the above: the L conditioner projections (wavenet.py:108), executed once per utterance."""
    return 2.0 * c["residual_layers"] * c["d_encoder"] * 2 * c["residual_channels"]


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


def e2e_flops(frames_total, n_steps, samples_total, n_utt_frames_hoist, h=NSF_V1, denoise=True):
    """(algorithmic, executed) FLOPs of one bench step.  Algorithmic = the reference's op count (every step pays the
    conditioner projections).  Executed = what the device ran: the conditioner projections once per utterance."""
    voc = nsf_flops_per_sample(h) * samples_total
    if not denoise:
        return voc, voc
    alg = wavenet_flops_per_frame() * frames_total * n_steps + voc
    return alg, alg - wavenet_hoisted_flops_per_frame() * n_utt_frames_hoist * (n_steps - 1)


# ====================================================================================================== modules and inputs
def seeded_modules(device, seed=1234, nsf=None, denoiser=True):
    """Random-init weights of the named architecture (no checkpoints exist offline).  The reference zero-inits the
    final projection (wavenet.py:192) and N(0,0.01)-inits the vocoder, which would make every activation ~0: use
    fan-in scaled draws so the data flowing through the kernels has O(1) magnitude (DVFS sees realistic toggling)."""
    from fish_diffusion_amd import DIFFUSIONS, NsfHifiGAN
    from fish_diffusion_amd.nsf_hifigan import generator_param_table
    nsf = nsf or NSF_V1
    torch.manual_seed(seed)
    diff = None
    if denoiser:
        diff = DIFFUSIONS.build(dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", **WN_CFG),
                                     spec_min=[-5], spec_max=[0], sampler_interval=10))
        torch.nn.init.normal_(diff.denoise_fn.output_projection.conv.weight, std=0.02)
        diff = diff.to(device).eval()
    g = torch.Generator().manual_seed(seed + 1)
    state = {}
    for key, shape, _ in generator_param_table(nsf):
        if key.endswith("bias") or len(shape) < 3:
            state[key] = torch.randn(shape, generator=g) * 0.01
        else:
            fan_in = shape[1] * shape[2] if "ups." not in key else shape[0] * shape[2] / max(1, nsf["upsample_rates"][int(key.split(".")[1])])
            state[key] = torch.randn(shape, generator=g) * math.sqrt(1.0 / max(1.0, fan_in))
    voc = NsfHifiGAN.from_state(nsf, state, use_natural_log=False)
    return diff, voc.to(device).eval()


def synth_f0(T, frame_rate=44100 / 512):
    """220 * 2^(0.3 sin(2 pi 0.7 t)) Hz with frames 100-130 unvoiced."""
    t = torch.arange(T, dtype=torch.float32) / frame_rate
    f0 = 220.0 * torch.pow(2.0, 0.3 * torch.sin(2 * math.pi * 0.7 * t))
    f0[100:130] = 0.0
    return f0


def synth_inputs(B, T, device, seed):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, 256, generator=g)
    return feats.to(device), synth_f0(T)[None].repeat(B, 1).contiguous().to(device)


def one_step(diff, voc, feats, f0, interval, streams=None):
    """One utterance batch: sampler, then vocoder (kept for tools/*: the headline config's step)."""
    mel = diff(feats, sampler_interval=interval)                       # [B, T, M] (log10-scale mel, diff_svc_v2)
    return voc.model(mel.transpose(1, 2), f0, mel_scale=2.30259)     # spec2wav for a batch (nsf_hifigan.py:72-85)


# ====================================================================================================== helpers
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


TRAFFIC_KEYS = {"convgate": ("EpiGate",), "outproj": ("EpiResSkip",), "nsf_resblock": ("2, false, 1, EpiResblock",)}


def pmc_traffic(config: str, kernel: str, expect: dict):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_traffic.py -> profiles/*_pmc_traffic.json;
    FETCH_SIZE and WRITE_SIZE need separate passes, so bench.py cannot collect them itself).  A file is only used for the
    workload it was collected on: its "workload" record must equal `expect` (files without one are the round-1 headline files:
    batch 1, T = 861)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        wl = d.get("workload", {"config": "headline", "batch": 1, "frames": 861})
        if wl != expect:
            continue
        for k, v in d["kernels"].items():
            if any(s in k for s in TRAFFIC_KEYS[kernel]):
                return v["hbm_bytes"], os.path.relpath(path, ROOT)
    return None, None


def prof_begin(handle, kind, stride):
    from fish_diffusion_amd import _lib
    _lib.check(_lib.lib().fdx_prof_select(handle.h, kind), handle.h)
    _lib.check(_lib.lib().fdx_prof_enable(handle.h, stride), handle.h)


def prof_pause(handle):
    from fish_diffusion_amd import _lib
    _lib.check(_lib.lib().fdx_prof_enable(handle.h, -1), handle.h)


def prof_end(handle):
    """(launches, avg_ms, flops_per_launch, label) of the launches recorded since prof_begin.  `label` is the library's own
    description of the kernel instantiation those launches ran (fdx_prof_label) -- never a literal in this file."""
    from fish_diffusion_amd import _lib
    n, ms, fl = C.c_int(), C.c_double(), C.c_double()
    buf = C.create_string_buffer(256)
    _lib.check(_lib.lib().fdx_prof_label(handle.h, buf, len(buf)), handle.h)
    _lib.check(_lib.lib().fdx_prof_read(handle.h, C.byref(n), C.byref(ms), C.byref(fl)), handle.h)
    _lib.check(_lib.lib().fdx_prof_enable(handle.h, 0), handle.h)
    if not n.value:
        return 0, 0.0, 0.0, ""
    return n.value, ms.value / n.value, fl.value, buf.value.decode()


def roofline_entry(kernel_desc, n, avg_ms, flops, peak, sampling, traffic=None, traffic_src=None, alg_bytes=None):
    ach = flops / (avg_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": kernel_desc, "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)",
            "traffic_source": traffic_src, "algorithmic_bytes": alg_bytes,
            "hbm_fraction": (round(traffic / (avg_ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4) if traffic else None),
            "peak_note": "nominal peak at the 2.4 GHz boost clock; tools/ktrace.py (s_memtime vs s_memrealtime, instrumented build) measured the fp32 "
                         "residual-block kernels at 2.05-2.19 GHz on this chip, i.e. an attainable matrix roof of ~134 TFLOP/s (profiles/r03_ktrace_headline_fp32.txt)",
            "launches_timed": n, "sampling": sampling, "avg_launch_us": round(avg_ms * 1e3, 2),
            "timing": "hipExtLaunchKernel start/stop events on the launch stream, timed region", "flops_per_launch": flops}


def cpu_denoiser(diff):
    from oracle import wavenet_ref
    sd = {k: v.detach().cpu() for k, v in diff.denoise_fn.state_dict().items()}
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
    hop = nsf["hop_size"]
    runs = []
    with torch.no_grad():
        den = cpu_denoiser(diff) if diff is not None else None
        feats, x0 = torch.randn(1, T, 256, generator=g), torch.randn(1, 128, T, generator=g)
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
            t0 = time.perf_counter()
            nsf_hifigan_ref.generator_forward(gsd, nsf, melv, f0, ri, sn)
            t_voc = time.perf_counter() - t0
            if not warm:
                runs.append((t_den, t_voc))
    med = sorted(runs, key=lambda p: p[0] + p[1])[len(runs) // 2]
    return med[0], med[1], cores, runs


def flush_c_stdio():
    try:
        C.CDLL(None).fflush(None)   # RCCL prints its version banner through C stdio: keep the JSON line the LAST line of stdout
    except Exception:
        pass


# ====================================================================================================== launcher dry run
def dry_run(args, cfg, steps, warmup, rank, world):
    """`--dry-run`: the multi-rank plumbing of this file on CPU ranks over gloo -- rendezvous, one broadcast of a byte arena from
    rank 0 (what `broadcast_model_weights` does with the packed weights), barrier-bracketed timing of a stand-in step, MAX over
    ranks, per-rank stats gather, rank 0's JSON line.  Not a measurement: the line says so."""
    from fish_diffusion_amd import dist as fdist
    import torch.distributed as tdist
    cpu = torch.device("cpu")
    t0 = time.perf_counter()
    arena = torch.arange(1 << 16, dtype=torch.int64).to(torch.uint8) if rank == 0 else torch.zeros(1 << 16, dtype=torch.uint8)
    if tdist.is_initialized():
        tdist.broadcast(arena, src=0)
    assert int(arena[259]) == 3, "arena broadcast failed"
    t_weights = time.perf_counter() - t0
    lens = torch.randint(516, 862, (64,), generator=torch.Generator().manual_seed(4)).tolist()
    mine = fdist.shard_utterances(lens, rank, world) if cfg == "sharded" else [0]
    frames = sum(lens[i] for i in mine) if cfg == "sharded" else 861
    audio_s = frames * 512 / 44100.0
    a = torch.randn(64, 64, generator=torch.Generator().manual_seed(rank))

    def step(k):
        return (a @ a).sum()

    def barrier():
        if tdist.is_initialized():
            tdist.barrier()
    for k in range(warmup):
        step(k)
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    t_local = time.perf_counter()
    barrier()
    dt = fdist.barrier_max(time.perf_counter() - t0, cpu)
    per_rank = fdist.gather_stats([(t_local - t0) / steps * 1e3, audio_s, float(frames), float(len(mine))], cpu)
    audio_all = fdist.sum_over_ranks(audio_s, cpu)
    out = {"metric": "DRY RUN (launcher / collective plumbing on CPU ranks; not a measurement)", "value": round(steps * audio_all / dt, 3),
           "unit": "audio-seconds/sec", "n_gpus": tdist.get_world_size() if tdist.is_initialized() else 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if (cfg == "sharded" and world > 1) else "weak",
           "vs_baseline": None, "dtype": "none", "data": "none", "dry_run": True, "backend": tdist.get_backend() if tdist.is_initialized() else None,
           "config": {"workload": "stand-in step function", "name": cfg, "parallelism": f"utterance-sharded x{world} (no per-step collective)"},
           "launched_by": os.environ.get("FDX_LAUNCHED_BY", "external launcher" if "WORLD_SIZE" in os.environ else "single process"),
           "weights_pack_bcast_s": round(t_weights, 4), "rccl_ranks": 0,
           "per_rank_ms": [round(float(v), 4) for v in per_rank[:, 0]], "per_rank_frames": [int(v) for v in per_rank[:, 2]],
           "per_rank_utterances": [int(v) for v in per_rank[:, 3]], "roofline": None, "cpu_baseline": None}
    if tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


# ====================================================================================================== main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="headline", help="headline | vocoder | sharded | ddpm1000 (aliases: 1..4, c1 c2 c3 c5)")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (headline: 1, vocoder: 32, ddpm1000: 16)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--interval", type=int, default=None, help="sampler_interval (headline / sharded: 10 => 100 UniPC steps; ddpm1000: 1)")
    ap.add_argument("--virtual-world", type=int, default=8, help="sharded config, single process: play rank 0 of this many ranks (1 = all 64 utterances)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-steps", type=int, default=None, help="denoiser calls the CPU baseline actually runs (the rest extrapolated linearly)")
    ap.add_argument("--no-prof", action="store_true", help="do not time the dominant kernel with HIP events")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive repeat of the headline step")
    ap.add_argument("--storage", choices=("fp32", "bf16", "fp16x3"), default="fp32",
                    help="bf16: the WaveNet's OPT-IN bf16 storage mode (BASELINE configs[4]); not parity-grade, reported as its own dtype.  "
                         "fp16x3: the OPT-IN fp16-split mode (hi + lo operands, three fp16 MFMAs per product block, fp32 accumulate): fp32-class "
                         "results (64 x 64 tiles below 200 wide tiles, 128-wide LDS tiles above); reported as its own dtype")
    ap.add_argument("--prof-stride", type=int, default=None, help="time every N-th launch of the dominant kernel")
    ap.add_argument("--no-exact", action="store_true", help="sharded config: the reference's padded-batch semantics (x_masks / cond_masks) instead of "
                    "the library's exact-ragged batches (every utterance as if run alone; padding tiles skipped)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / collective plumbing only: CPU ranks over gloo, a stand-in step function "
                    "(tests/test_bench_host.py runs this at world size 2; never a measurement)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` starts its own N ranks (one process per GPU, RCCL), like the reference's sharded tool spawns its
        # own workers (tools/preprocessing/extract_features.py:262-322).  Under torch.distributed.run WORLD_SIZE is set and we are a rank.
        from fish_diffusion_amd import dist as fdist
        sys.stdout.flush()
        raise SystemExit(fdist.launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], need_gpus=not args.dry_run))
    cfg = ALIASES.get(str(args.config).lower())
    if cfg is None:
        raise SystemExit(f"unknown --config {args.config!r}")
    steps, warmup = DEFAULT_STEPS[cfg]
    steps = args.steps if args.steps is not None else steps
    warmup = args.warmup if args.warmup is not None else warmup

    from fish_diffusion_amd import _lib, dist as fdist, pipeline

    rank, local_rank, world = fdist.init_process_group("gloo" if args.dry_run else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, cfg, steps, warmup, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but this node exposes {torch.cuda.device_count()} GPU(s)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    def sync_barrier():
        """synchronize + barrier + synchronize; returns the time this rank reached the barrier (its own work done)."""
        torch.cuda.synchronize()
        t_local = time.perf_counter()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()
        return t_local

    nsf = NSF_V1_256 if cfg == "vocoder" else NSF_V1
    hop = nsf["hop_size"]
    T = int(args.seconds * 44100) // hop
    diff, voc = seeded_modules(dev, nsf=nsf, denoiser=cfg != "vocoder")
    # rank 0's packed weights reach the other ranks by one RCCL broadcast (outside the timed region)
    t0 = time.perf_counter()
    fdist.broadcast_model_weights(diff.denoise_fn if diff is not None else None, voc.model, dev, src=0)
    torch.cuda.synchronize()
    t_weights = time.perf_counter() - t0
    if args.storage != "fp32":
        if diff is None:
            raise SystemExit(f"--storage {args.storage} applies to the denoiser")
        diff.denoise_fn.storage = args.storage
    voc.model.rng = "philox"          # perf mode: source noise drawn on the device inside the library
    bf16 = args.storage == "bf16"
    f16s = args.storage == "fp16x3"
    # fp16x3: an fp32-class product block costs three fp16 MFMAs -> the roof for ALGORITHMIC flops is a third of the fp16 MFMA peak
    peak = PEAK_BF16_TFLOPS if bf16 else (round(PEAK_BF16_TFLOPS / 3.0, 1) if f16s else PEAK_F32_TFLOPS)

    n_total = steps + warmup
    extra = {}
    # ------------------------------------------------------------------------------------------------ per-config step
    if cfg == "headline":
        B = args.batch or 1
        interval = args.interval or 10
        n_steps = 1000 // interval
        pool = [synth_inputs(B, T, dev, 1234 + rank + 1000 * k)[0] for k in range(n_total)]   # a fresh conditioner per step
        f0 = synth_inputs(B, T, dev, 0)[1]

        def step(k):
            return one_step(diff, voc, pool[k], f0, interval)
        audio_s = B * T * hop / 44100.0
        alg, exe = e2e_flops(B * T, n_steps, B * T * hop, B * T, nsf)
        metric = BASELINE_METRIC if n_steps == 100 else f"audio-seconds/sec/GPU ({n_steps}-step denoise + NSF-HiFiGAN, 44.1 kHz)"
        workload = (f"BASELINE configs[1]: svc_hubert_soft (diff_svc_v2 WaveNet C=512 x 20 layers) {n_steps}-step UniPC + NSF-HiFiGAN "
                    f"config_v1 (hop 512), batch={B} x {args.seconds:g} s @44.1 kHz (T={T}) per GPU, fresh features every step")
        cfg_extra = {"batch_per_gpu": B, "frames": T, "sampler": "unipc", "sampler_steps": n_steps}
        prof_handle = lambda: diff.denoise_fn.engine(dev)   # noqa: E731
        prof_kind, stride = _lib.PROF_WN_CONVGATE, args.prof_stride or 7   # 7 is co-prime with the 20 layers: every dilation sampled
        C_, M_ = WN_CFG["residual_channels"], B * T
        alg_bytes = 4 * (2 * C_ * 3 * C_ + C_ * M_ + 2 * C_ * M_ + C_ * M_)   # weights + Y in + conditioner slab in + Z out
        kwhat = "dilated conv k=3 + gate of the residual block"
        traffic_key, traffic_expect = "convgate", {"config": "headline" + ("_bf16" if bf16 else "_fp16x3" if f16s else ""), "batch": B, "frames": T}
    elif cfg == "vocoder":
        B = args.batch or 32
        n_steps = 0
        g = torch.Generator().manual_seed(2000 + rank)
        mels = [(torch.randn(B, 128, T, generator=g) * 0.5 - 2.0).to(dev) for _ in range(2)]
        f0 = synth_f0(T, 44100 / hop)[None].repeat(B, 1).contiguous().to(dev)

        def step(k):
            return voc.model(mels[k & 1], f0)
        audio_s = B * T * hop / 44100.0
        alg, exe = e2e_flops(0, 0, B * T * hop, 0, nsf, denoise=False)
        metric = "audio-seconds/sec/GPU (NSF-HiFiGAN vocoder only, 44.1 kHz)"
        workload = (f"BASELINE configs[2]: NSF-HiFiGAN only, tools/nsf_hifigan/config_v1_256.json (hop 256), batch={B} x {args.seconds:g} s mel "
                    f"(T={T}) per GPU")
        cfg_extra = {"batch_per_gpu": B, "frames": T, "hop": hop}
        prof_handle = lambda: voc.model.engine(dev)   # noqa: E731
        prof_kind, stride = _lib.PROF_NSF_RESBLOCK, args.prof_stride or 5
        alg_bytes = None
        kwhat = ("the ResBlock1 convs (k = 3/7/11, leaky-relu on the operand, residual / MRF mean in the epilogue) of the stages with >= 64 "
                 "channels; FLOP-weighted over the launches timed")
        traffic_key, traffic_expect = "nsf_resblock", {"config": "vocoder", "batch": B, "frames": T}
    elif cfg == "sharded":
        interval = args.interval or 10
        n_steps = 1000 // interval
        vworld = world if world > 1 else max(1, args.virtual_world)
        vrank = rank if world > 1 else 0
        g = torch.Generator().manual_seed(4)
        lens = torch.randint(516, 862, (64,), generator=g).tolist()       # 6-10 s at hop 512
        feats = [torch.randn(n, 256, generator=g).to(dev) for n in lens]
        f0s = [synth_f0(n).to(dev) for n in lens]
        mine = fdist.shard_utterances(lens, vrank, vworld)
        batches = pipeline.make_batches([lens[i] for i in mine], 8, padding_free=not args.no_exact)

        def step(k):
            return pipeline.synthesize(diff, voc, feats, f0s, max_batch=8, sampler_interval=interval, rank=vrank, world=vworld,
                                       exact=not args.no_exact)
        frames = sum(lens[i] for i in mine)
        audio_s = frames * hop / 44100.0
        alg, exe = e2e_flops(frames, n_steps, frames * hop, frames, nsf)
        B = max(len(b) for b in batches)
        metric = "audio-seconds/sec/GPU (100-step denoise + NSF-HiFiGAN, 44.1 kHz; 64 ragged utterances sharded by utterance)"
        workload = (f"BASELINE configs[3]: svc_content_vec, 64 utterances of 6-10 s (T in [516, 861]) sharded longest-first over {vworld} ranks"
                    + ("" if world > 1 else f" (this process = rank 0 of a virtual {vworld}-way job)")
                    + f"; this rank: {len(mine)} utterances, {frames} frames, masked micro-batches {[len(b) for b in batches]}; {n_steps}-step UniPC + "
                    "NSF-HiFiGAN config_v1 per utterance")
        cfg_extra = {"utterances_total": 64, "utterances_this_rank": len(mine), "frames_this_rank": frames, "shards": vworld,
                     "micro_batches": [len(b) for b in batches], "sampler": "unipc", "sampler_steps": n_steps,
                     "batching": "reference padded-batch semantics (x_masks / cond_masks)" if args.no_exact else
                                 "exact-ragged (utterances laid end to end in one row with 16-frame holes: no padding to a common length; every utterance "
                                 "bit-identical to its batch-1 run)"}
        prof_handle = lambda: diff.denoise_fn.engine(dev)   # noqa: E731
        prof_kind, stride = _lib.PROF_WN_CONVGATE, args.prof_stride or 7
        alg_bytes = None
        kwhat = "dilated conv k=3 + gate of the residual block (micro-batches)" + ("; peak = fp16 MFMA peak / 3" if f16s else "")
        traffic_key, traffic_expect = "convgate", {"config": "sharded" + ("_fp16x3" if f16s else ""), "batch": B, "frames": max(lens[i] for i in mine)}
    else:  # ddpm1000
        from fish_diffusion_amd import DiffSinger, pitch_to_scale
        B = args.batch or 16
        interval = args.interval or 1
        n_steps = 1000 // interval
        mcfg = dict(text_encoder=dict(type="NaiveProjectionEncoder", input_size=256, output_size=256),
                    speaker_encoder=dict(type="NaiveProjectionEncoder", input_size=128, output_size=256, use_embedding=True),
                    pitch_encoder=dict(type="NaiveProjectionEncoder", input_size=1, output_size=256, preprocessing=pitch_to_scale),
                    diffusion=dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", **WN_CFG), spec_min=[-5], spec_max=[0]))
        torch.manual_seed(77)
        model = DiffSinger(mcfg).to(dev).eval()
        model.diffusion = diff                           # the seeded full-size denoiser
        diff.step_rng = "philox"                         # per-step noise from the device generator (no [1000, B, M, T] tensor)
        g = torch.Generator().manual_seed(5 + rank)
        contents = [torch.randn(B, T, 256, generator=g).to(dev) for _ in range(2)]
        f0 = synth_inputs(B, T, dev, 0)[1]
        spk = torch.randint(0, 128, (B,), generator=g).to(dev)

        def step(k):
            mel = model.infer(spk, contents[k & 1], f0, sampler_interval=interval, noise_predictor="naive")
            return voc.model(mel.transpose(1, 2).contiguous(), f0, mel_scale=2.30259)
        audio_s = B * T * hop / 44100.0
        alg, exe = e2e_flops(B * T, n_steps, B * T * hop, B * T, nsf)
        metric = f"audio-seconds/sec/GPU ({n_steps}-step DDPM denoise + NSF-HiFiGAN, 44.1 kHz)"
        workload = (f"BASELINE configs[4] as SURVEY F4 reads it: diff_svc_v2 WaveNet, DDPM (naive) sampler, {n_steps} denoiser calls, multi-speaker "
                    f"front end (128-entry speaker embedding), batch={B} x {args.seconds:g} s per GPU (= batch 128 over 8 GPUs), then NSF-HiFiGAN config_v1; "
                    + ("bf16 storage / fp32 accumulate (opt-in mode)" if bf16 else
                       "fp16-split operands (hi + lo), 3 fp16 MFMAs per product block, fp32 accumulate (opt-in mode, fp32-class)" if f16s else "fp32"))
        cfg_extra = {"batch_per_gpu": B, "frames": T, "sampler": "naive (DDPM ancestral)", "sampler_steps": n_steps, "step_noise": "device Philox"}
        prof_handle = lambda: diff.denoise_fn.engine(dev)   # noqa: E731
        prof_kind, stride = _lib.PROF_WN_CONVGATE, args.prof_stride or 97   # co-prime with 20: every layer sampled, ~200 launches
        C_, M_ = WN_CFG["residual_channels"], B * T
        esz = 2 if bf16 else 4      # (fp16x3: hi + lo = 4 bytes per element)
        alg_bytes = esz * (2 * C_ * 3 * C_ + C_ * M_ + C_ * M_) + 4 * 2 * C_ * M_    # weights + Y in + Z out (+ fp32 conditioner slab)
        kwhat = f"dilated conv k=3 + gate of the residual block at batch {B}" + ("; hi.lo + lo.hi + hi.hi, peak = fp16 MFMA peak / 3" if f16s else "")
        traffic_key, traffic_expect = "convgate", {"config": "ddpm1000" + ("_bf16" if bf16 else "_fp16x3" if f16s else ""), "batch": B, "frames": T}

    # ------------------------------------------------------------------------------------------------ warm-up, timed region
    for k in range(warmup):
        step(k)
    sync_barrier()
    do_prof = not args.no_prof
    if do_prof:
        prof_begin(prof_handle(), prof_kind, stride)
        sync_barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        out = step(warmup + k)
        if k == 0 and do_prof:       # the dominant kernel is timed on the first timed step only (it needs the eager launch path);
            prof_pause(prof_handle())   # the other steps replay the hipGraph
    t_local = sync_barrier()
    dt = time.perf_counter() - t0
    per_rank = fdist.gather_stats([(t_local - t0) / steps * 1e3, audio_s, float(cfg_extra.get("frames_this_rank", B * T)),
                                   float(cfg_extra.get("utterances_this_rank", B))], dev)   # [world, 4]
    dt = fdist.barrier_max(dt, dev)
    del out

    roofline = None
    if do_prof:
        n, avg_ms, fl, label = prof_end(prof_handle())
        if n:
            traffic, traffic_src = pmc_traffic(cfg, traffic_key, traffic_expect)
            roofline = roofline_entry(f"{label}: {kwhat}", n, avg_ms, fl, peak, f"every {stride}th launch of the first timed step", traffic, traffic_src,
                                      alg_bytes)

    # ------------------------------------------------------------------------------------------------ outside the timed region
    other = []
    stages = None
    if cfg in ("headline", "ddpm1000", "sharded") and do_prof and rank == 0:   # the second residual-block kernel, one extra step
        prof_begin(prof_handle(), _lib.PROF_WN_OUTPROJ, stride)
        step(warmup)
        torch.cuda.synchronize()
        n, avg_ms, fl, label = prof_end(prof_handle())
        if n:
            tr, src = pmc_traffic(cfg, "outproj", traffic_expect)
            C_, M_ = WN_CFG["residual_channels"], fl / (2.0 * 2 * WN_CFG["residual_channels"] ** 2)   # columns per launch, from its flops
            esz = 2 if bf16 else 4
            # weights [2C x C] + Z in + X in/out + SK in/out + next layer's Y out (fp32 residual stream in every mode)
            ob = esz * (2 * C_ * C_ + C_ * M_) + 4 * (4 * C_ * M_) + esz * C_ * M_
            e = roofline_entry(f"{label}: 1x1 out-projection + residual / skip epilogue" + (" (HBM-bound: the fp32 residual stream and skip sum "
                               "are read and written every layer)" if bf16 else ""), n,
                               avg_ms, fl, peak, f"every {stride}th launch of one extra step outside the timed region", tr, src, int(ob))
            if bf16:
                e["bound"] = "hbm"
            other.append(e)
    if cfg == "headline":
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        mel = diff(pool[0], sampler_interval=interval)
        ev[1].record()
        voc.model(mel.transpose(1, 2), f0, mel_scale=2.30259)
        ev[2].record()
        torch.cuda.synchronize()
        stages = {"denoise": round(ev[0].elapsed_time(ev[1]), 2), "vocoder": round(ev[1].elapsed_time(ev[2]), 2)}
        if not args.no_pcie:      # the same step with host-resident inputs / outputs: reported beside `value`
            hf = [p.cpu().pin_memory() for p in pool[:max(2, min(len(pool), steps))]]
            hf0 = f0.cpu().pin_memory()
            hw = torch.empty((B, 1, T * hop), dtype=torch.float32).pin_memory()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(steps):
                w = one_step(diff, voc, hf[k % len(hf)].to(dev, non_blocking=True), hf0.to(dev, non_blocking=True), interval)
                hw.copy_(w, non_blocking=True)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t1
            extra["pcie_inclusive"] = {"value": round(steps * audio_s / dtp, 3), "ms_per_step": round(dtp / steps * 1e3, 3),
                                       "bytes_h2d_per_step": int(hf[0].numel() * 4 + hf0.numel() * 4), "bytes_d2h_per_step": int(hw.numel() * 4),
                                       "note": "features + f0 start in pinned host memory, waveform ends there; measured on this rank after the timed "
                                               "region -- reported beside `value`, never as `value`"}

    audio_all = fdist.sum_over_ranks(audio_s, dev)      # weak configs: world x audio_s; sharded: the ranks' shards differ
    value = steps * audio_all / dt
    alg, exe = fdist.sum_over_ranks(alg, dev) / world, fdist.sum_over_ranks(exe, dev) / world   # per-GPU means
    e2e_alg = alg * steps / dt / 1e12
    e2e_exe = exe * steps / dt / 1e12
    out = {
        "metric": metric, "value": round(value, 3), "unit": "audio-seconds/sec",
        "n_gpus": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if (cfg == "sharded" and world > 1) else "weak", "vs_baseline": None,
        "dtype": ("bf16 storage / f32 accumulate (opt-in mode, not parity-grade)" if bf16 else
                  "fp16 hi+lo split operands x3 MFMA / f32 accumulate (opt-in mode, fp32-class: held to the fp32 parity bars)" if f16s
                  else "f32"),
        "data": "synthetic (seeded N(0,1) features, vibrato f0 with an unvoiced gap; random-init weights of the named architecture)",
        "config": dict({"workload": workload, "name": cfg,
                        "parallelism": f"utterance-sharded x{world} (no per-step collective)"}, **cfg_extra),
        "value_is": "whole-job aggregate over n_gpus (per_gpu = value / n_gpus); inputs resident in HBM",
        "per_gpu": round(value / world, 3), "x_realtime_per_gpu": round(value / world, 3),
        "stages_ms": stages,
        "end_to_end": {"tflops": round(e2e_alg, 3), "frac_of_peak": round(e2e_alg / peak, 4),
                       "tflops_executed": round(e2e_exe, 3), "frac_of_peak_executed": round(e2e_exe / peak, 4), "peak_tflops": peak,
                       "algorithmic_flops_per_step": alg, "executed_flops_per_step": exe,
                       "note": "algorithmic = the reference's op count (SURVEY 8d); executed = what the device ran (the step-invariant "
                               "conditioner projections once per utterance instead of once per sampler step)"},
        "weights_pack_upload_s" if not torch.distributed.is_initialized() else "weights_pack_bcast_s": round(t_weights, 4),
        "launched_by": os.environ.get("FDX_LAUNCHED_BY", "external launcher" if "WORLD_SIZE" in os.environ else "single process"),
        "rccl_ranks": (torch.distributed.get_world_size() if torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" else 0),
        "per_rank_ms": [round(float(v), 3) for v in per_rank[:, 0]],
        "per_rank_audio_s": [round(float(v), 3) for v in per_rank[:, 1]],
        "per_rank_frames": [int(v) for v in per_rank[:, 2]],
        "per_rank_utterances": [int(v) for v in per_rank[:, 3]],
        "imbalance": round(float(per_rank[:, 0].max() / per_rank[:, 0].mean()), 4),
        "roofline": roofline,
        "other_kernels": other or None,
    }
    out.update(extra)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if cfg == "headline":
            ss = args.cpu_sample_steps or 100
            td, tv, cores, runs = cpu_chain(diff, voc, nsf, T, n_steps, ss)
            sample = (f"1 x {args.seconds:g} s utterance (T={T}): {ss} of {n_steps} UniPC steps timed ({td / n_steps * 1e3:.0f} ms/step"
                      + ("" if ss == n_steps else f", extrapolated x{n_steps / ss:g}") + f") + full NSF-HiFiGAN pass ({tv:.2f} s)")
            cpu_audio = args.seconds
        elif cfg == "vocoder":
            td, tv, cores, runs = cpu_chain(None, voc, nsf, T, 0, 0)
            sample = f"1 of the {B} x {args.seconds:g} s mels (T={T}): one full NSF-HiFiGAN config_v1_256 pass ({tv:.2f} s)"
            cpu_audio = args.seconds
        elif cfg == "sharded":
            ss = args.cpu_sample_steps or 20
            Tm = sorted(lens[i] for i in mine)[len(mine) // 2]
            td, tv, cores, runs = cpu_chain(diff, voc, nsf, Tm, n_steps, ss)
            sample = (f"1 utterance of median length (T={Tm}) run alone: {ss} of {n_steps} UniPC steps timed, extrapolated x{n_steps / ss:g}, + full "
                      f"NSF-HiFiGAN pass ({tv:.2f} s)")
            cpu_audio = Tm * hop / 44100.0
        else:
            ss = args.cpu_sample_steps or 50
            td, tv, cores, runs = cpu_chain(diff, voc, nsf, T, n_steps, ss, predictor="naive")
            sample = (f"1 of the {B} x {args.seconds:g} s utterances (T={T}): {ss} of {n_steps} DDPM steps timed ({td / n_steps * 1e3:.0f} ms/step, "
                      f"extrapolated x{n_steps / ss:g}) + full NSF-HiFiGAN pass ({tv:.2f} s); fp32")
            cpu_audio = args.seconds
        cb = {"value": round(cpu_audio / (td + tv), 4), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
              "sample": sample + f"; torch {torch.__version__} CPU, {cores} threads; 1 warm-up pass + median of {len(runs)} timed passes",
              "denoise_s": round(td, 4), "vocoder_s": round(tv, 4), "protocol": "BASELINE.md section 3: 1 warm-up + 3 timed, median",
              "runs_s": [[round(a, 4), round(b, 4)] for a, b in runs],
              "runs_value": [round(cpu_audio / (a + b), 4) for a, b in runs]}
        out["cpu_baseline"] = cb
        out["gpu_over_cpu"] = round(value / cb["value"], 1)
    else:
        out["cpu_baseline"] = None
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        flush_c_stdio()
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
