#!/usr/bin/env python3
"""bench.py -- the hot path's metrics on MI355X, one JSON line per run.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--no-extras]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--config` selects what `value` is measured on (default: the config the headline metric is quoted on):

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
  SURVEY 8(f) rows (the callers / model families either side of the path):
  hifisinger_v2   what configs/svc_hifisinger_v2.py REALLY runs (archs/hifisinger/core.py:115-141): NaiveProjection encoders ->
            feature_fuser -> RefineGANGenerator (refinegan/generator.py:437-478, num_mels = hidden 256, hop 256), 16 x 10 s per GPU.
  convnext  ConvNextDenoiser (modules/convnext.py:155-262, dim 512 x 20 blocks) under the 100-step UniPC sampler, batch 1 x 10 s.
  tfdec     TransformerDecoderDenoiser (modules/convnext.py:263-379, dim 512 x 12 layers) under the same sampler, batch 1 x 10 s.

THE DEFAULT RUN (`python bench.py`, one GPU) measures the headline config as `value` and then, in the same process, every other
BASELINE config and every SURVEY 8(f) row as a short timed run of its own: `"configs": {"vocoder", "sharded", "ddpm1000"}` and
`"widening": {"hifisinger_v2", "convnext", "tfdec"}`, each with ms_per_step, x_realtime_per_gpu, end_to_end.frac_of_peak, dtype and the
roofline of its dominant kernel from the library's own launch-stream events (`fdx_prof_*`).  `--no-extras` skips them.

Inputs are resident in HBM when the timed region starts; `value` is the whole-job aggregate over all ranks (weak scaling: every
rank runs the same per-GPU workload on its own utterances; the only collective is the start-up RCCL broadcast of the packed
weights, outside the timed region, and the MAX over ranks of the wall time).  `pcie_inclusive` (headline only) repeats the step
with features / f0 starting in pinned host memory and the waveform copied back: reported beside `value`, never as `value`.

`roofline` is for the config's dominant kernel: algorithmic FLOPs per launch / average launch duration from HIP events recorded
on the launch stream inside the timed region (`fdx_prof_*`), against the fp32 MFMA roof (157.3 TFLOP/s: the path is fp32 for
parity and compute-bound); `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes of the same
command (profiles/).  `clock_mhz` = the shader clock the driver reports (sysfs pp_dpm_sclk of this GPU) sampled every 20 ms while
the timed region runs.  `cpu_baseline` = the CPU oracle (pinned restatement of the reference's PyTorch path, torch CPU ops) timed
on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchkit.cpu import cpu_baseline_leg, cpu_chain, usable_cores  # noqa: E402,F401
from benchkit.flops import *  # noqa: E402,F401,F403  (model tables and FLOP formulas: tests and tools read them from here)
from benchkit.timing import (SclkSampler, first_call_entry, measure, other_kernel, pmc_traffic, prof_begin, prof_end, prof_pause,  # noqa: E402,F401
                             roofline_entry)
from benchkit.workloads import build_work, one_step, seeded_modules, seeded_refinegan, synth_f0, synth_inputs  # noqa: E402,F401

ALIASES = {"headline": "headline", "1": "headline", "c1": "headline", "configs1": "headline",
           "vocoder": "vocoder", "2": "vocoder", "c2": "vocoder", "configs2": "vocoder",
           "sharded": "sharded", "3": "sharded", "c3": "sharded", "c4": "sharded", "configs3": "sharded",
           "ddpm1000": "ddpm1000", "4": "ddpm1000", "c5": "ddpm1000", "configs4": "ddpm1000",
           "hifisinger_v2": "hifisinger_v2", "hifisinger": "hifisinger_v2", "convnext": "convnext", "tfdec": "tfdec"}
DEFAULT_STEPS = {"headline": (5, 2), "vocoder": (5, 2), "sharded": (3, 1), "ddpm1000": (2, 1),
                 "hifisinger_v2": (5, 2), "convnext": (5, 2), "tfdec": (3, 1)}
# the short runs the default line carries beside `value` (steps, warm-up passes); ddpm1000's warm-up is one 10-step pass of the same shapes
EXTRA_CONFIGS = {"vocoder": (3, 1), "sharded": (2, 1), "ddpm1000": (1, 1)}
EXTRA_WIDENING = {"hifisinger_v2": (3, 1), "convnext": (3, 1), "tfdec": (2, 1)}


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




ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes", "launches_timed", "sampling",
                 "avg_launch_us", "flops_per_launch")


def compact(w, m, cpu=None):
    """A sub-result of the default line: the same accounting as the main line, without the per-rank legs; `cpu` = its bounded CPU leg."""
    r = m.roofline
    return {"cpu_baseline": None if cpu is None else cpu[0], "gpu_over_cpu": None if cpu is None else cpu[1], "first_call_ms": first_call_entry(m, w),
            "metric": w.metric, "value": round(m.value, 3), "unit": "audio-seconds/sec", "x_realtime_per_gpu": round(m.value / m.world, 3),
            "steps": m.steps, "warmup": m.warmup, "ms_per_step": round(m.dt / m.steps * 1e3, 3), "dtype": w.dtype,
            "warmup_note": ("one 10-step pass of the same shapes (allocation, module load), then ONE timed 1000-step pass" if w.warm else None),
            "workload": w.workload, "config": dict({"name": w.name}, **w.cfg_extra),
            "end_to_end": {"tflops": round(m.e2e_alg, 3), "frac_of_peak": round(m.e2e_alg / w.peak, 4), "tflops_executed": round(m.e2e_exe, 3),
                           "frac_of_peak_executed": round(m.e2e_exe / w.peak, 4), "peak_tflops": w.peak, "algorithmic_flops_per_step": m.alg},
            "roofline": (None if r is None else {k: r[k] for k in ROOFLINE_KEYS})}


def release(w):
    """Drop a workload's modules and device buffers before the next one is built (the library frees its arenas with the handle)."""
    import gc
    for k in list(vars(w)):
        setattr(w, k, None)
    gc.collect()
    torch.cuda.empty_cache()


def headline_stages_and_pcie(w, steps, args, dev, extra):
    """Per-stage times of one more step (torch events on the current stream: whole stages, not single kernels) and the PCIe-inclusive repeat."""
    diff, voc, pool, f0, interval = w.diff, w.voc, w.pool, w.f0, w.interval
    B, T, hop = w.B, w.T, w.hop
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    voc.wav2spec(torch.zeros(B, T * hop, device=dev))      # (first call: DFT / filterbank tables, frame buffers)
    ev[0].record()
    mel = diff(pool[0], sampler_interval=interval)
    ev[1].record()
    wav = voc.model(mel.transpose(1, 2), f0, mel_scale=2.30259)
    ev[2].record()
    # SURVEY 8(d)'s fourth stage: the STFT / mel front end (`NsfHifiGAN.wav2spec`, nsf_hifigan.py:91-107) on the waveform just produced --
    # outside the timed region (the mel -> waveform path does not call it), here for the per-stage figure only
    voc.wav2spec(wav[:, 0])
    ev[3].record()
    torch.cuda.synchronize()
    stages = {"denoise": round(ev[0].elapsed_time(ev[1]), 2), "vocoder": round(ev[1].elapsed_time(ev[2]), 2),
              "mel": round(ev[2].elapsed_time(ev[3]), 3),
              "note": "torch events around whole stages of ONE extra step after the timed region; `mel` = wav2spec (reflect pad + Hann + DFT + magnitude + "
                      "slaney filterbank + log) of the produced waveform, not part of the mel -> waveform step"}
    if not args.no_pcie:      # the same step with host-resident inputs / outputs: reported beside `value`
        hf = [p.cpu().pin_memory() for p in pool[:max(2, min(len(pool), steps))]]
        hf0 = f0.cpu().pin_memory()
        hw = torch.empty((B, 1, T * hop), dtype=torch.float32).pin_memory()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(steps):
            wv = one_step(diff, voc, hf[k % len(hf)].to(dev, non_blocking=True), hf0.to(dev, non_blocking=True), interval)
            hw.copy_(wv, non_blocking=True)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t1
        extra["pcie_inclusive"] = {"value": round(steps * w.audio_s / dtp, 3), "ms_per_step": round(dtp / steps * 1e3, 3),
                                   "bytes_h2d_per_step": int(hf[0].numel() * 4 + hf0.numel() * 4), "bytes_d2h_per_step": int(hw.numel() * 4),
                                   "note": "features + f0 start in pinned host memory, waveform ends there; measured on this rank after the timed "
                                           "region -- reported beside `value`, never as `value`"}
    return stages


# ====================================================================================================== main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="headline", help="headline | vocoder | sharded | ddpm1000 (aliases: 1..4, c1 c2 c3 c5) | hifisinger_v2 | convnext | tfdec")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (headline: 1, vocoder: 32, ddpm1000 / hifisinger_v2: 16)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--interval", type=int, default=None, help="sampler_interval (headline / sharded: 10 => 100 UniPC steps; ddpm1000: 1)")
    ap.add_argument("--virtual-world", type=int, default=8, help="sharded config, single process: play rank 0 of this many ranks (1 = all 64 utterances)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="default line only: skip the short runs of the other BASELINE configs / SURVEY 8(f) rows")
    ap.add_argument("--extras", default=None, help="comma-separated subset of the extras to run (vocoder,sharded,ddpm1000,hifisinger_v2,convnext,tfdec)")
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

    from fish_diffusion_amd import dist as fdist

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
    if args.storage != "fp32" and cfg not in ("headline", "sharded", "ddpm1000"):
        raise SystemExit(f"--storage {args.storage} applies to the WaveNet denoiser configs")

    do_prof = not args.no_prof
    w = build_work(cfg, args, dev, rank, world, steps + warmup)
    m = measure(w, steps, warmup, args, dev, do_prof, sclk=True, prof_outside=cfg in ("convnext", "tfdec"))
    hop, T, B, n_steps, nsf, peak = w.hop, w.T, w.B, w.n_steps, w.nsf, w.peak
    extra = {}

    # ------------------------------------------------------------------------------------------------ outside the timed region
    other = []
    stages = None
    if w.other_prof is not None and do_prof and rank == 0:
        e = other_kernel(w, w.other_prof, args, warmup)
        if e:
            other.append(e)
    if cfg == "headline":
        stages = headline_stages_and_pcie(w, steps, args, dev, extra)

    value, dt, per_rank = m.value, m.dt, m.per_rank
    out = {
        "metric": w.metric, "value": round(value, 3), "unit": "audio-seconds/sec",
        "n_gpus": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True,
        "scaling": w.scaling, "vs_baseline": None,
        "dtype": w.dtype,
        "data": "synthetic (seeded N(0,1) features, vibrato f0 with an unvoiced gap; random-init weights of the named architecture)",
        "config": dict({"workload": w.workload, "name": cfg,
                        "parallelism": f"utterance-sharded x{world} (no per-step collective)"}, **w.cfg_extra),
        "value_is": "whole-job aggregate over n_gpus (per_gpu = value / n_gpus); inputs resident in HBM",
        "per_gpu": round(value / world, 3), "x_realtime_per_gpu": round(value / world, 3),
        "stages_ms": stages,
        "end_to_end": {"tflops": round(m.e2e_alg, 3), "frac_of_peak": round(m.e2e_alg / peak, 4),
                       "tflops_executed": round(m.e2e_exe, 3), "frac_of_peak_executed": round(m.e2e_exe / peak, 4), "peak_tflops": peak,
                       "algorithmic_flops_per_step": m.alg, "executed_flops_per_step": m.exe,
                       "note": "algorithmic = the reference's op count (SURVEY 8d); executed = what the device ran (the step-invariant "
                               "conditioner projections once per utterance instead of once per sampler step)"},
        "weights_pack_upload_s" if not torch.distributed.is_initialized() else "weights_pack_bcast_s": round(w.t_weights, 4),
        "launched_by": os.environ.get("FDX_LAUNCHED_BY", "external launcher" if "WORLD_SIZE" in os.environ else "single process"),
        "rccl_ranks": (torch.distributed.get_world_size() if torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" else 0),
        "per_rank_ms": [round(float(v), 3) for v in per_rank[:, 0]],
        "per_rank_audio_s": [round(float(v), 3) for v in per_rank[:, 1]],
        "per_rank_frames": [int(v) for v in per_rank[:, 2]],
        "per_rank_utterances": [int(v) for v in per_rank[:, 3]],
        "imbalance": round(float(per_rank[:, 0].max() / per_rank[:, 0].mean()), 4),
        "clock_mhz": m.clock,
        "first_call_ms": first_call_entry(m, w),
        "roofline": m.roofline,
        "other_kernels": other or None,
    }
    out.update(extra)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["gpu_over_cpu"] = cpu_baseline_leg(w, cfg, args, value)
    else:
        out["cpu_baseline"] = None

    # ------------------------------------------------------------------------------------------------ the other configs / rows, one GPU
    default_line = cfg == "headline" and world == 1 and args.storage == "fp32" and args.batch is None and args.interval is None and args.seconds == 10.0
    if default_line and not args.no_extras:
        release(w)
        only = None if args.extras is None else {ALIASES.get(s.strip().lower(), s.strip().lower()) for s in args.extras.split(",") if s.strip()}
        t_extras = time.perf_counter()
        for key, table in (("configs", EXTRA_CONFIGS), ("widening", EXTRA_WIDENING)):
            out[key] = {}
            for name, (st, wu) in table.items():
                if only is not None and name not in only:
                    continue
                t1 = time.perf_counter()
                try:
                    we = build_work(name, args, dev, rank, world, st + wu, extra=True)
                    me = measure(we, st, wu, args, dev, do_prof, prof_outside=name in ("convnext", "tfdec"))
                    cpu = None
                    if rank == 0 and not args.no_cpu_baseline:      # every published line carries its CPU figure (bounded: a few seconds each)
                        try:
                            cpu = cpu_baseline_leg(we, name, args, me.value, quick=True)
                        except Exception as e:   # noqa: BLE001
                            cpu = ({"error": f"{type(e).__name__}: {e}"}, None)
                    res = compact(we, me, cpu)
                    release(we)
                except Exception as e:   # noqa: BLE001  (a failed sub-run must not cost the headline line; it is reported as what it is)
                    res = {"error": f"{type(e).__name__}: {e}"}
                res["wall_s_incl_setup"] = round(time.perf_counter() - t1, 2)
                out[key][name] = res
        out["extras_wall_s"] = round(time.perf_counter() - t_extras, 2)
        out["extras_note"] = ("each entry is its own barrier-bracketed timed run in this process after the headline's (same build, same GPU, fp32, inputs "
                              "resident in HBM); `python bench.py --config <name>` gives the same workload as `value` with more steps and the CPU leg")
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        flush_c_stdio()
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
