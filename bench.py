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
import glob
import json
import math
import os
import sys
import threading
import time
from types import SimpleNamespace

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
CN_CFG = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=20)   # modules/convnext.py:156-166 defaults
TD_CFG = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=12)   # modules/convnext.py:264-272 defaults
RG_HIFISINGER = dict(sampling_rate=44100, hop_length=256, downsample_rates=[2, 2, 8, 8], upsample_rates=[8, 8, 2, 2],
                     leaky_relu_slope=0.2, num_mels=256, start_channels=16)   # configs/_base_/archs/hifi_svc_v2.py:43-52
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 matrix == vector peak; tools/ubench/mfmaclk.hip measures 155.1 on this part
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0
BASELINE_METRIC = "audio-seconds/sec/GPU (100-step denoise + NSF-HiFiGAN, 44.1 kHz)"   # BASELINE.json "metric", verbatim

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


def refinegan_flops(T, cfg=RG_HIFISINGER):
    """2*MAC of every Conv1d in RefineGANGenerator.forward for ONE item of T frames (refinegan/generator.py:333-423: template_conv,
    the down path's ResBlocks, mel_conv, source_conv, per up stage input_conv + 3 ParallelResBlock branches of 6 convs, output_conv) --
    the quantity torch.utils.flop_counter reports on the reference module (tools/flops_reference.py checks the formula against it)."""
    c, L = cfg["start_channels"], T * cfg["hop_length"]
    fl = 2.0 * c * 7 * L
    length = L
    for r in cfg["downsample_rates"]:
        length //= r
        fl += 2.0 * length * 7 * (2 * c * c + 5 * (2 * c) ** 2)
        c *= 2
    fl += 2.0 * T * 7 * cfg["num_mels"] * c
    c *= 2
    sf0 = 1
    for r in cfg["upsample_rates"][1:]:
        sf0 *= r
    fl += 2.0 * (T * cfg["upsample_rates"][0]) * c * 2 * sf0
    length = T
    for r in cfg["upsample_rates"]:
        length *= r
        n = c // 2
        fl += 2.0 * length * (7 * (c + c // 4) * n + sum(6 * k * n * n for k in (3, 7, 11)))
        c = n
    fl += 2.0 * L * 7 * c
    return fl


def hifisinger_frontend_flops(T, content_dim=768, hidden=256):
    """text Linear + the two feature_fuser Linears (archs/hifisinger/core.py:24-29,70-107); the scalar encoders are O(hidden) per frame."""
    return 2.0 * T * (content_dim * hidden + 2 * hidden * hidden)


def convnext_flops_per_frame(c=CN_CFG):
    """(algorithmic, hoisted) per frame per denoiser call: 2*MAC of every conv / linear of ConvNext.forward (modules/convnext.py:206-262):
    input_projection, conditioner_projection (2 convs), per block condition_projection + depthwise k=7 + pwconv1 + pwconv2, output_projection.
    Hoisted = what the device runs once per utterance instead of once per call (the conditioner MLP and the L condition projections)."""
    M, D, E, L = c["mel_channels"], c["dim"], c["condition_dim"], c["num_layers"]
    H = D * c["mlp_factor"]
    hoist = 2.0 * (E * H + H * D + L * D * D)
    return 2.0 * (M * D + L * (7 * D + 2 * D * H) + D * D + D * M) + hoist, hoist


def tfdec_flops_per_frame(T, c=TD_CFG):
    """(algorithmic, hoisted) per frame per call of TransformerDecoderDenoiser.forward (modules/convnext.py:330-379) at T frames: the 1x1 conv
    projections, per nn.TransformerDecoderLayer the self-attention (in_proj 3 D^2, QK^T + PV = 4 T D, out_proj D^2), the cross-attention
    (q D^2, k / v of the memory 2 D^2, QK^T + PV, out_proj) and the feed-forward (2 D H).  Hoisted: condition_projection (step-invariant)."""
    M, D, E, L = c["mel_channels"], c["dim"], c["condition_dim"], c["num_layers"]
    H = D * c["mlp_factor"]
    hoist = 2.0 * (E * H + H * D)
    gemm = 2.0 * (M * H + H * D + L * (3 * D * D + D * D + D * D + 2 * D * D + D * D + 2 * D * H) + D * D + D * M)
    attn = L * 2 * 4.0 * T * D
    return gemm + attn + hoist, hoist


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


def seeded_refinegan(cfg, seed=9):
    """A RefineGANGenerator with fan-in scaled weights (rgbench's recipe): O(1) activations through every conv."""
    from fish_diffusion_amd import RefineGANGenerator
    torch.manual_seed(seed)
    gen = RefineGANGenerator(**cfg)
    gen.remove_weight_norm()
    with torch.no_grad():
        for k, p in gen.named_parameters():
            if p.dim() == 3:
                p.copy_(torch.randn_like(p) * (1.0 / (p.shape[1] * p.shape[2])) ** 0.5)
            elif k.endswith("weight"):
                p.fill_(0.1)
    return gen


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


TRAFFIC_KEYS = {"convgate": ("EpiGate",), "outproj": ("EpiResSkip",), "nsf_resblock": ("2, false, 1, EpiResblock",),
                "rg_resblock": ("2, false, 1, EpiResblock",), "cn_pwconv1": ("2, true, 2, EpiBias",), "td_attn": ("k_attn_qs",)}


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
    buf = C.create_string_buffer(320)
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
            "peak_note": "nominal fp32 matrix peak at the 2.4 GHz boost clock.  tools/ubench/mfmaclk.hip on this part (profiles/r04_mfma_clock_ubench.txt): an "
                         "MFMA-only v_mfma_f32_16x16x4_f32 loop on 256 CUs sustains 155.1 TFLOP/s at 2.39 GHz, as one long launch and as a chain of 12 / 25 us "
                         "launches alike; with the residual-block K loop's load mix (6 dwordx4 per 16 MFMAs from L2) 116.5 at the SAME 2.39 GHz: operand delivery "
                         "bounds the K loop.  The library's own kernels, same counters (s_memtime / s_memrealtime per wave, instrumented build, last two stamps "
                         "taken back to back: profiles/r05_ktrace_headline_fp32_adjacent_stamps.txt), read 2.10 (conv + gate) / 2.18 GHz (out-projection) while sclk "
                         "reports 2.38-2.40 at ~1100 W of board power (`clock_mhz`).  Round 5 ruled out wait states (a wave that only sleeps reads 2.397 GHz), barriers, "
                         "LDS reductions, exp phases, cold-load waits, combined L2 + LDS + MFMA load (all 2.38-2.39, profiles/r05_clock_*_ubench.txt) and the stamps "
                         "themselves; no cause is named (profiles/NOTES.md round 5 item 4) -- `peak` stays the nominal 157.3",
            "launches_timed": n, "sampling": sampling, "avg_launch_us": round(avg_ms * 1e3, 2),
            "timing": "hipExtLaunchKernel start/stop events on the launch stream", "flops_per_launch": flops}


class SclkSampler:
    """Shader clock of THIS GPU as the driver reports it (sysfs pp_dpm_sclk: the level marked '*'), sampled from a thread while the
    timed region runs.  The card is matched by PCI address (torch's device properties); no match -> no samples (reported as such)."""

    def __init__(self, dev_index: int, period_s: float = 0.02):
        self.period, self.samples, self._stop, self._thr, self.path, self.why = period_s, [], threading.Event(), None, None, None
        self.power_path, self.power = None, []      # board power (hwmon, microwatts) beside the clock: VERDICT r4 item 4(a)
        try:
            p = torch.cuda.get_device_properties(dev_index)
            want = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            for card in sorted(glob.glob("/sys/class/drm/card*/device")):
                if os.path.basename(os.path.realpath(card)) == want and os.path.exists(os.path.join(card, "pp_dpm_sclk")):
                    self.path = os.path.join(card, "pp_dpm_sclk")
                    pw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input")))
                    self.power_path = pw[0] if pw else None
            if self.path is None:
                self.why = f"no /sys/class/drm/card*/device matches PCI {want}"
        except Exception as e:   # noqa: BLE001
            self.why = f"{type(e).__name__}: {e}"

    def _read(self):
        try:
            for ln in open(self.path).read().splitlines():
                if ln.rstrip().endswith("*"):
                    return float(ln.split(":")[1].strip().split("M")[0])
        except Exception:
            return None
        return None

    def _run(self):
        while not self._stop.is_set():
            v = self._read()
            if v is not None:
                self.samples.append(v)
            if self.power_path:
                try:
                    self.power.append(float(open(self.power_path).read()) / 1e6)
                except Exception:   # noqa: BLE001
                    pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self.path:
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thr:
            self._thr.join(timeout=1.0)

    def report(self):
        if not self.samples:
            return {"mean": None, "samples": 0, "source": self.path, "note": self.why or "no samples"}
        s = sorted(self.samples)
        out = {"mean": round(sum(s) / len(s), 1), "median": s[len(s) // 2], "min": s[0], "max": s[-1], "samples": len(s),
               "period_ms": self.period * 1e3, "source": self.path,
               "note": "sysfs pp_dpm_sclk ('*' level) of this GPU, sampled by a host thread during the timed region"}
        if self.power:
            out["board_power_w"] = {"mean": round(sum(self.power) / len(self.power), 1), "max": round(max(self.power), 1), "samples": len(self.power),
                                    "source": self.power_path}
        return out


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


# ====================================================================================================== the workloads
def build_work(cfg, args, dev, rank, world, n_total, extra=False):
    """Everything a config needs to be stepped and accounted for: modules (random-init weights of the named architecture), inputs resident
    in HBM, the step function, the algorithmic / executed FLOPs of a step, what `roofline` times.  `extra`: the short run the default line
    carries beside `value` (same workload, same accounting; ddpm1000 gets a 10-step warm-up pass instead of a 1000-step one)."""
    from fish_diffusion_amd import _lib, dist as fdist, pipeline
    storage = "fp32" if extra else args.storage
    bf16, f16s = storage == "bf16", storage == "fp16x3"
    seconds = 10.0 if extra else args.seconds
    nsf = NSF_V1_256 if cfg == "vocoder" else NSF_V1
    hop = RG_HIFISINGER["hop_length"] if cfg == "hifisinger_v2" else nsf["hop_size"]
    T = int(seconds * 44100) // hop
    # fp16x3: an fp32-class product block costs three fp16 MFMAs -> the roof for ALGORITHMIC flops is a third of the fp16 MFMA peak
    peak = PEAK_BF16_TFLOPS if bf16 else (round(PEAK_BF16_TFLOPS / 3.0, 1) if f16s else PEAK_F32_TFLOPS)
    w = SimpleNamespace(name=cfg, T=T, hop=hop, nsf=nsf, peak=peak, bf16=bf16, f16s=f16s, storage=storage, seconds=seconds, warm=None,
                        diff=None, voc=None, t_weights=0.0, alg_bytes=None, lens=None, mine=None, scaling="weak", other_prof=None,
                        dtype=("bf16 storage / f32 accumulate (opt-in mode, not parity-grade)" if bf16 else
                               "fp16 hi+lo split operands x3 MFMA / f32 accumulate (opt-in mode, fp32-class: held to the fp32 parity bars)" if f16s
                               else "f32"))
    batch = None if extra else args.batch
    interval_arg = None if extra else args.interval

    if cfg in ("headline", "vocoder", "sharded", "ddpm1000"):
        diff, voc = seeded_modules(dev, nsf=nsf, denoiser=cfg != "vocoder")
        # rank 0's packed weights reach the other ranks by one RCCL broadcast (outside the timed region)
        t0 = time.perf_counter()
        fdist.broadcast_model_weights(diff.denoise_fn if diff is not None else None, voc.model, dev, src=0)
        torch.cuda.synchronize()
        w.t_weights = time.perf_counter() - t0
        if storage != "fp32":
            if diff is None:
                raise SystemExit(f"--storage {storage} applies to the denoiser")
            diff.denoise_fn.storage = storage
        voc.model.rng = "philox"          # perf mode: source noise drawn on the device inside the library
        w.diff, w.voc = diff, voc

    if cfg == "headline":
        B = batch or 1
        interval = interval_arg or 10
        n_steps = 1000 // interval
        pool = [synth_inputs(B, T, dev, 1234 + rank + 1000 * k)[0] for k in range(n_total)]   # a fresh conditioner per step
        f0 = synth_inputs(B, T, dev, 0)[1]
        w.pool, w.f0, w.interval = pool, f0, interval
        w.step = lambda k: one_step(diff, voc, pool[k % len(pool)], f0, interval)
        w.audio_s = B * T * hop / 44100.0
        w.alg, w.exe = e2e_flops(B * T, n_steps, B * T * hop, B * T, nsf)
        w.metric = BASELINE_METRIC if n_steps == 100 else f"audio-seconds/sec/GPU ({n_steps}-step denoise + NSF-HiFiGAN, 44.1 kHz)"
        w.workload = (f"BASELINE configs[1]: svc_hubert_soft (diff_svc_v2 WaveNet C=512 x 20 layers) {n_steps}-step UniPC + NSF-HiFiGAN "
                      f"config_v1 (hop 512), batch={B} x {seconds:g} s @44.1 kHz (T={T}) per GPU, fresh features every step")
        w.cfg_extra = {"batch_per_gpu": B, "frames": T, "sampler": "unipc", "sampler_steps": n_steps}
        w.prof_handle = lambda: diff.denoise_fn.engine(dev)
        w.prof_kind, w.stride = _lib.PROF_WN_CONVGATE, args.prof_stride or 7   # 7 is co-prime with the 20 layers: every dilation sampled
        C_, M_ = WN_CFG["residual_channels"], B * T
        w.alg_bytes = 4 * (2 * C_ * 3 * C_ + C_ * M_ + 2 * C_ * M_ + C_ * M_)   # weights + Y in + conditioner slab in + Z out
        w.kwhat = "dilated conv k=3 + gate of the residual block"
        w.traffic_key, w.traffic_expect = "convgate", {"config": "headline" + ("_bf16" if bf16 else "_fp16x3" if f16s else ""), "batch": B, "frames": T}
        w.other_prof = _lib.PROF_WN_OUTPROJ
    elif cfg == "vocoder":
        B = batch or 32
        n_steps = 0
        g = torch.Generator().manual_seed(2000 + rank)
        mels = [(torch.randn(B, 128, T, generator=g) * 0.5 - 2.0).to(dev) for _ in range(2)]
        f0 = synth_f0(T, 44100 / hop)[None].repeat(B, 1).contiguous().to(dev)
        w.step = lambda k: voc.model(mels[k & 1], f0)
        w.audio_s = B * T * hop / 44100.0
        w.alg, w.exe = e2e_flops(0, 0, B * T * hop, 0, nsf, denoise=False)
        w.metric = "audio-seconds/sec/GPU (NSF-HiFiGAN vocoder only, 44.1 kHz)"
        w.workload = (f"BASELINE configs[2]: NSF-HiFiGAN only, tools/nsf_hifigan/config_v1_256.json (hop 256), batch={B} x {seconds:g} s mel "
                      f"(T={T}) per GPU")
        w.cfg_extra = {"batch_per_gpu": B, "frames": T, "hop": hop}
        w.prof_handle = lambda: voc.model.engine(dev)
        w.prof_kind, w.stride = _lib.PROF_NSF_RESBLOCK, args.prof_stride or 5
        w.kwhat = ("the ResBlock1 convs (k = 3/7/11, leaky-relu on the operand, residual / MRF mean in the epilogue) of the stages with >= 64 "
                   "channels; FLOP-weighted over the launches timed")
        w.traffic_key, w.traffic_expect = "nsf_resblock", {"config": "vocoder", "batch": B, "frames": T}
    elif cfg == "sharded":
        interval = interval_arg or 10
        n_steps = 1000 // interval
        vworld = world if world > 1 else max(1, args.virtual_world)
        vrank = rank if world > 1 else 0
        exact = extra or not args.no_exact
        g = torch.Generator().manual_seed(4)
        lens = torch.randint(516, 862, (64,), generator=g).tolist()       # 6-10 s at hop 512
        feats = [torch.randn(n, 256, generator=g).to(dev) for n in lens]
        f0s = [synth_f0(n).to(dev) for n in lens]
        mine = fdist.shard_utterances(lens, vrank, vworld)
        batches = pipeline.make_batches([lens[i] for i in mine], 8, padding_free=exact)
        w.failures = []      # (utterance id, reason) of this rank: the serving loop isolates failures per utterance (pipeline.synthesize on_error)
        w.step = lambda k: pipeline.synthesize(diff, voc, feats, f0s, max_batch=8, sampler_interval=interval, rank=vrank, world=vworld,
                                               exact=exact, on_error="isolate", failures=w.failures)
        frames = sum(lens[i] for i in mine)
        w.lens, w.mine = lens, mine
        w.audio_s = frames * hop / 44100.0
        w.alg, w.exe = e2e_flops(frames, n_steps, frames * hop, frames, nsf)
        B = max(len(b) for b in batches)
        w.scaling = "strong" if world > 1 else "weak"
        w.metric = "audio-seconds/sec/GPU (100-step denoise + NSF-HiFiGAN, 44.1 kHz; 64 ragged utterances sharded by utterance)"
        w.workload = (f"BASELINE configs[3]: svc_content_vec, 64 utterances of 6-10 s (T in [516, 861]) sharded longest-first over {vworld} ranks"
                      + ("" if world > 1 else f" (this process = rank 0 of a virtual {vworld}-way job)")
                      + f"; this rank: {len(mine)} utterances, {frames} frames, masked micro-batches {[len(b) for b in batches]}; {n_steps}-step UniPC + "
                      "NSF-HiFiGAN config_v1 per utterance")
        w.cfg_extra = {"utterances_total": 64, "utterances_this_rank": len(mine), "frames_this_rank": frames, "shards": vworld,
                       "micro_batches": [len(b) for b in batches], "sampler": "unipc", "sampler_steps": n_steps,
                       "batching": "exact-ragged (utterances laid end to end in one row with 16-frame holes: no padding to a common length; every utterance "
                                   "bit-identical to its batch-1 run)" if exact else "reference padded-batch semantics (x_masks / cond_masks)"}
        w.prof_handle = lambda: diff.denoise_fn.engine(dev)
        w.prof_kind, w.stride = _lib.PROF_WN_CONVGATE, args.prof_stride or 7
        w.kwhat = "dilated conv k=3 + gate of the residual block (micro-batches)" + ("; peak = fp16 MFMA peak / 3" if f16s else "")
        w.traffic_key, w.traffic_expect = "convgate", {"config": "sharded" + ("_fp16x3" if f16s else ""), "batch": B, "frames": max(lens[i] for i in mine)}
        w.other_prof = _lib.PROF_WN_OUTPROJ
    elif cfg == "ddpm1000":
        from fish_diffusion_amd import DiffSinger, pitch_to_scale
        B = batch or 16
        interval = interval_arg or 1
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

        def step(k, iv=interval):
            mel = model.infer(spk, contents[k & 1], f0, sampler_interval=iv, noise_predictor="naive")
            return voc.model(mel.transpose(1, 2).contiguous(), f0, mel_scale=2.30259)
        w.step = step
        if extra:   # a 10-step pass of the same shapes: allocations, module load, first-touch -- then ONE timed 1000-step pass
            w.warm = lambda k: step(k, 100)
        w.audio_s = B * T * hop / 44100.0
        w.alg, w.exe = e2e_flops(B * T, n_steps, B * T * hop, B * T, nsf)
        w.metric = f"audio-seconds/sec/GPU ({n_steps}-step DDPM denoise + NSF-HiFiGAN, 44.1 kHz)"
        w.workload = (f"BASELINE configs[4] as SURVEY F4 reads it: diff_svc_v2 WaveNet, DDPM (naive) sampler, {n_steps} denoiser calls, multi-speaker "
                      f"front end (128-entry speaker embedding), batch={B} x {seconds:g} s per GPU (= batch 128 over 8 GPUs), then NSF-HiFiGAN config_v1; "
                      + ("bf16 storage / fp32 accumulate (opt-in mode)" if bf16 else
                         "fp16-split operands (hi + lo), 3 fp16 MFMAs per product block, fp32 accumulate (opt-in mode, fp32-class)" if f16s else "fp32"))
        w.cfg_extra = {"batch_per_gpu": B, "frames": T, "sampler": "naive (DDPM ancestral)", "sampler_steps": n_steps, "step_noise": "device Philox"}
        w.prof_handle = lambda: diff.denoise_fn.engine(dev)
        w.prof_kind, w.stride = _lib.PROF_WN_CONVGATE, args.prof_stride or 97   # co-prime with 20: every layer sampled, ~200 launches
        C_, M_ = WN_CFG["residual_channels"], B * T
        esz = 2 if bf16 else 4      # (fp16x3: hi + lo = 4 bytes per element)
        w.alg_bytes = esz * (2 * C_ * 3 * C_ + C_ * M_ + C_ * M_) + 4 * 2 * C_ * M_    # weights + Y in + Z out (+ fp32 conditioner slab)
        w.kwhat = f"dilated conv k=3 + gate of the residual block at batch {B}" + ("; hi.lo + lo.hi + hi.hi, peak = fp16 MFMA peak / 3" if f16s else "")
        w.traffic_key, w.traffic_expect = "convgate", {"config": "ddpm1000" + ("_bf16" if bf16 else "_fp16x3" if f16s else ""), "batch": B, "frames": T}
        w.other_prof = _lib.PROF_WN_OUTPROJ
    elif cfg == "hifisinger_v2":
        from fish_diffusion_amd import HiFiSinger
        B = batch or 16
        n_steps = 0
        hid = RG_HIFISINGER["num_mels"]
        lin1 = dict(type="NaiveProjectionEncoder", input_size=1, output_size=hid)
        torch.manual_seed(31)
        model = HiFiSinger(dict(hidden_size=hid, text_encoder=dict(type="NaiveProjectionEncoder", input_size=768, output_size=hid),
                                speaker_encoder=dict(type="NaiveProjectionEncoder", input_size=10, output_size=hid, use_embedding=True),
                                pitch_shift_encoder=lin1, energy_encoder=lin1, encoder=dict(type="RefineGAN", **RG_HIFISINGER)))
        model.encoder = seeded_refinegan(RG_HIFISINGER)
        model = model.to(dev).eval()
        model.encoder.rng = "philox"
        g = torch.Generator().manual_seed(6 + rank)
        contents = [torch.randn(B, T, 768, generator=g).to(dev) for _ in range(2)]      # ContentVec features at the mel frame rate
        f0 = synth_f0(T, 44100 / hop)[None].repeat(B, 1).contiguous().to(dev)[:, :, None]
        lens = torch.full((B,), T, dtype=torch.long, device=dev)
        spk = torch.randint(0, 10, (B,), generator=g).to(dev)
        shift = torch.zeros(B, 1, device=dev)
        energy = (torch.rand(B, T, generator=g) * 0.1).to(dev)
        w.step = lambda k: model(spk, contents[k & 1], lens, T, pitches=f0, pitch_shift=shift, energy=energy)
        w.audio_s = B * T * hop / 44100.0
        w.alg = w.exe = B * (refinegan_flops(T, RG_HIFISINGER) + hifisinger_frontend_flops(T, 768, hid))
        w.metric = "audio-seconds/sec/GPU (HiFiSinger front end + RefineGAN generator, 44.1 kHz)"
        w.workload = (f"SURVEY 8(f) row 2 / what configs/svc_hifisinger_v2.py runs: NaiveProjection encoders (ContentVec 768 -> 256, 10 speakers, "
                      f"pitch-shift, energy) -> feature_fuser -> RefineGANGenerator (num_mels = 256, hop 256, start_channels 16), batch={B} x {seconds:g} s "
                      f"(T={T}) per GPU, device Philox noises")
        w.cfg_extra = {"batch_per_gpu": B, "frames": T, "hop": hop}
        w.prof_handle = lambda: model.encoder.engine(dev)
        w.prof_kind, w.stride = _lib.PROF_RG_RESBLOCK, args.prof_stride or 5
        w.kwhat = ("RefineGAN's ResBlock convs (k = 3/7/11 ParallelResBlock branches and the k = 7 down path; leaky-relu on the operand, residual in the "
                   "epilogue) on the one-tile-per-wave instantiation; FLOP-weighted over the launches timed")
        w.traffic_key, w.traffic_expect = "rg_resblock", {"config": "hifisinger_v2", "batch": B, "frames": T}
        w.keep = model
    elif cfg in ("convnext", "tfdec"):
        from fish_diffusion_amd import GaussianDiffusion
        from oracle import convnext_ref, tfdec_ref   # (seeded weights only: the same draws the parity tests use)
        B = batch or 1
        interval = interval_arg or 10
        n_steps = 1000 // interval
        mc = CN_CFG if cfg == "convnext" else TD_CFG
        diff = GaussianDiffusion(dict(type="ConvNextDenoiser" if cfg == "convnext" else "TransformerDecoderDenoiser", **mc), spec_min=[-5], spec_max=[0])
        diff.denoise_fn.load_state_dict((convnext_ref if cfg == "convnext" else tfdec_ref).seeded_state(1, **mc))
        diff = diff.to(dev).eval()
        pool = [synth_inputs(B, T, dev, 4321 + rank + 1000 * k)[0] for k in range(n_total)]
        w.step = lambda k: diff(pool[k % len(pool)], sampler_interval=interval)
        w.audio_s = B * T * hop / 44100.0
        per_frame, hoist = convnext_flops_per_frame(mc) if cfg == "convnext" else tfdec_flops_per_frame(T, mc)
        w.alg = per_frame * B * T * n_steps
        w.exe = w.alg - hoist * B * T * (n_steps - 1)
        w.metric = f"audio-seconds/sec/GPU ({n_steps}-step UniPC over the {'ConvNext' if cfg == 'convnext' else 'TransformerDecoder'} denoiser, mel only, 44.1 kHz / hop 512)"
        w.workload = (f"SURVEY 8(f) row 4: {'ConvNextDenoiser (dim 512 x 20 blocks, mlp 4)' if cfg == 'convnext' else 'TransformerDecoderDenoiser (dim 512 x 12 layers, 8 heads, mlp 4)'}"
                      f" behind the DENOISERS contract, {n_steps}-step UniPC, batch={B} x {seconds:g} s (T={T}), fresh features every step; features -> mel "
                      "(no vocoder pass)")
        w.cfg_extra = {"batch_per_gpu": B, "frames": T, "sampler": "unipc", "sampler_steps": n_steps}
        w.prof_handle = lambda: diff.denoise_fn.engine(dev)
        if cfg == "convnext":
            w.prof_kind, w.stride = _lib.PROF_CN_PWCONV1, args.prof_stride or 7
            w.kwhat = "pwconv1 (dim -> 4 dim) with the LayerNorm folded in and the GELU epilogue"
            w.traffic_key = "cn_pwconv1"
        else:
            w.prof_kind, w.stride = _lib.PROF_TD_ATTN, args.prof_stride or 7
            w.kwhat = "self- / cross-attention (QK^T + softmax + PV) of the decoder layers"
            w.traffic_key = "td_attn"
        w.traffic_expect = {"config": cfg, "batch": B, "frames": T}
        w.diff = diff
    else:
        raise SystemExit(f"unknown config {cfg!r}")
    w.B, w.n_steps = B, n_steps
    return w


def measure(w, steps, warmup, args, dev, do_prof, sclk=False, prof_outside=False):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by synchronize + barrier + synchronize on both sides; MAX over ranks.
    The dominant kernel is timed (launch-stream events) on the FIRST timed step only -- it needs the eager launch path; the other steps
    replay the recorded hipGraph.  `prof_outside` (the widening rows whose denoiser call is ~150 launches of 5-25 us: an eager step is bound by
    the host's launch rate, 250 ms against 205 for the transformer, and would be half of a 2-step timed region): the kernel is timed on ONE
    EXTRA step after the timed region instead, and every timed step replays the graph."""
    from fish_diffusion_amd import dist as fdist
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1

    def sync_barrier():
        torch.cuda.synchronize()
        t_local = time.perf_counter()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()
        return t_local

    for k in range(warmup):
        (w.warm or w.step)(k)
    sync_barrier()
    prof_in = do_prof and not prof_outside
    if prof_in:
        prof_begin(w.prof_handle(), w.prof_kind, w.stride)
        sync_barrier()
    sampler = SclkSampler(dev.index or 0) if sclk else None
    if sampler:
        sampler.__enter__()
    t0 = time.perf_counter()
    for k in range(steps):
        out = w.step(warmup + k)
        if k == 0 and prof_in:
            prof_pause(w.prof_handle())
    t_local = sync_barrier()
    dt = time.perf_counter() - t0
    if sampler:
        sampler.__exit__(None, None, None)
    per_rank = fdist.gather_stats([(t_local - t0) / steps * 1e3, w.audio_s, float(w.cfg_extra.get("frames_this_rank", w.B * w.T)),
                                   float(w.cfg_extra.get("utterances_this_rank", w.B))], dev)   # [world, 4]
    dt = fdist.barrier_max(dt, dev)
    if getattr(w, "failures", None) is not None:      # which utterances the job lost, over all ranks (none, on synthetic input)
        w.cfg_extra["failed_utterances"] = fdist.gather_failed(sorted({i for i, _ in w.failures}), dev)
    del out
    roofline = None
    if do_prof:
        if prof_outside:
            prof_begin(w.prof_handle(), w.prof_kind, w.stride)
            w.step(warmup + steps)
            torch.cuda.synchronize()
        n, avg_ms, fl, label = prof_end(w.prof_handle())
        if n:
            traffic, traffic_src = pmc_traffic(w.name, w.traffic_key, w.traffic_expect)
            where = "one extra step after the timed region" if prof_outside else "the first timed step"
            roofline = roofline_entry(f"{label}: {w.kwhat}", n, avg_ms, fl, w.peak, f"every {w.stride}th launch of {where}", traffic, traffic_src,
                                      w.alg_bytes)
    audio_all = fdist.sum_over_ranks(w.audio_s, dev)      # weak configs: world x audio_s; sharded: the ranks' shards differ
    alg, exe = fdist.sum_over_ranks(w.alg, dev) / world, fdist.sum_over_ranks(w.exe, dev) / world   # per-GPU means
    return SimpleNamespace(dt=dt, steps=steps, warmup=warmup, per_rank=per_rank, roofline=roofline, value=steps * audio_all / dt, alg=alg, exe=exe,
                           e2e_alg=alg * steps / dt / 1e12, e2e_exe=exe * steps / dt / 1e12, clock=sampler.report() if sampler else None, world=world)


def other_kernel(w, kind, args, steps_done):
    """The second residual-block kernel, timed on one extra step outside the timed region."""
    prof_begin(w.prof_handle(), kind, w.stride)
    w.step(steps_done)
    torch.cuda.synchronize()
    n, avg_ms, fl, label = prof_end(w.prof_handle())
    if not n:
        return None
    tr, src = pmc_traffic(w.name, "outproj", w.traffic_expect)
    C_, M_ = WN_CFG["residual_channels"], fl / (2.0 * 2 * WN_CFG["residual_channels"] ** 2)   # columns per launch, from its flops
    esz = 2 if w.bf16 else 4
    # weights [2C x C] + Z in + X in/out + SK in/out + next layer's Y out (fp32 residual stream in every mode)
    ob = esz * (2 * C_ * C_ + C_ * M_) + 4 * (4 * C_ * M_) + esz * C_ * M_
    e = roofline_entry(f"{label}: 1x1 out-projection + residual / skip epilogue" + (" (HBM-bound: the fp32 residual stream and skip sum "
                       "are read and written every layer)" if w.bf16 else ""), n,
                       avg_ms, fl, w.peak, f"every {w.stride}th launch of one extra step outside the timed region", tr, src, int(ob))
    if w.bf16:
        e["bound"] = "hbm"
    return e


def compact(w, m):
    """A sub-result of the default line: the same accounting as the main line, without the per-rank / CPU legs."""
    r = m.roofline
    return {"metric": w.metric, "value": round(m.value, 3), "unit": "audio-seconds/sec", "x_realtime_per_gpu": round(m.value / m.world, 3),
            "steps": m.steps, "warmup": m.warmup, "ms_per_step": round(m.dt / m.steps * 1e3, 3), "dtype": w.dtype,
            "warmup_note": ("one 10-step pass of the same shapes (allocation, module load), then ONE timed 1000-step pass" if w.warm else None),
            "workload": w.workload, "config": dict({"name": w.name}, **w.cfg_extra),
            "end_to_end": {"tflops": round(m.e2e_alg, 3), "frac_of_peak": round(m.e2e_alg / w.peak, 4), "tflops_executed": round(m.e2e_exe, 3),
                           "frac_of_peak_executed": round(m.e2e_exe / w.peak, 4), "peak_tflops": w.peak, "algorithmic_flops_per_step": m.alg},
            "roofline": (None if r is None else {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                                    "launches_timed", "sampling", "avg_launch_us", "flops_per_launch")})}


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


def cpu_baseline_leg(w, cfg, args, value):
    """The oracle chain timed on this box's host cores on a bounded sample of the same workload (rank 0, one GPU)."""
    diff, voc, nsf, T, n_steps, B, hop = w.diff, w.voc, w.nsf, w.T, w.n_steps, w.B, w.hop
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
        Tm = sorted(w.lens[i] for i in w.mine)[len(w.mine) // 2]
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
    return cb, round(value / cb["value"], 1)


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
        "roofline": m.roofline,
        "other_kernels": other or None,
    }
    out.update(extra)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and cfg in ("headline", "vocoder", "sharded", "ddpm1000"):
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
                    res = compact(we, me)
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
