"""Workload construction for bench.py: random-init modules of the named architectures, synthetic inputs resident in HBM, the step
function of every configuration and its accounting (what `value` counts, the FLOPs of a step, what `roofline` times)."""
from __future__ import annotations

import math
import time
from types import SimpleNamespace

import torch

from .flops import *  # noqa: F401,F403  (model tables + formulas)
from .flops import (BASELINE_METRIC, CN_CFG, NSF_V1, NSF_V1_256, PEAK_BF16_TFLOPS, PEAK_F32_TFLOPS, RG_HIFISINGER, TD_CFG, WN_CFG, attention_bytes,
                    convgate_bytes, convnext_flops_per_frame, e2e_flops, hifisinger_frontend_flops, nsf_resblock_bytes, pwconv1_bytes,
                    refinegan_flops, refinegan_resblock_bytes, tfdec_flops_per_frame)


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


def seeded_denoiser(kind, cfg, seed=1):
    """A ConvNextDenoiser / TransformerDecoderDenoiser behind GaussianDiffusion with random-init weights of the architecture: the modules' own
    initialisation (the reference's: fan-in bounds, xavier in-projections) under a fixed seed, with the parameters the reference starts at
    constants -- LayerNorm affine, ConvNext's layer scale (1e-6: every block would be the identity), attention biases -- drawn as well so that
    O(1) data flows through every kernel.  (Round 5 borrowed the parity tests' seeded state from oracle/: the bench no longer imports it here.)"""
    from fish_diffusion_amd import GaussianDiffusion
    torch.manual_seed(seed)
    diff = GaussianDiffusion(dict(type=kind, **cfg), spec_min=[-5], spec_max=[0])
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for k, p in diff.denoise_fn.named_parameters():
            if k.endswith("gamma"):
                p.copy_(0.1 + 0.02 * torch.randn(p.shape, generator=g))
            elif "norm" in k and k.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif ("norm" in k and k.endswith("bias")) or k.endswith("in_proj_bias") or k.endswith("out_proj.bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return diff


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
        w.alg_bytes = convgate_bytes(B * T)    # weights + Y in + conditioner slab in + Z out
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
        w.alg_bytes = int(nsf_resblock_bytes(T, B, nsf))      # mean over the launches of the family (weights + every operand / result once)
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
        # mean over the micro-batches (each runs the same number of launches); an exact-ragged micro-batch is ONE row: its items + 16-frame holes
        rows = [sum(lens[mine[j]] for j in b) + 16 * (len(b) - 1) if exact else len(b) * max(lens[mine[j]] for j in b) for b in batches]
        w.alg_bytes = int(sum(convgate_bytes(n, esz=4) for n in rows) / len(rows))
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
        w.alg_bytes = convgate_bytes(B * T, esz=2 if bf16 else 4)    # weights + Y in + Z out (+ fp32 conditioner slab); fp16x3: hi + lo = 4 bytes
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
        w.alg_bytes = int(refinegan_resblock_bytes(T, B, RG_HIFISINGER))
        w.keep = model
    elif cfg in ("convnext", "tfdec"):
        B = batch or 1
        interval = interval_arg or 10
        n_steps = 1000 // interval
        mc = CN_CFG if cfg == "convnext" else TD_CFG
        diff = seeded_denoiser("ConvNextDenoiser" if cfg == "convnext" else "TransformerDecoderDenoiser", mc).to(dev).eval()
        pool = [synth_inputs(B, T, dev, 4321 + rank + 1000 * k)[0] for k in range(n_total)]
        w.step = lambda k: diff(pool[k % len(pool)], sampler_interval=interval)
        w.audio_s = B * T * hop / 44100.0
        per_frame, hoist = convnext_flops_per_frame(mc) if cfg == "convnext" else tfdec_flops_per_frame(T, mc)
        w.alg = per_frame * B * T * n_steps
        w.exe = w.alg - hoist * B * T * (n_steps - 1)
        fam = "ConvNext" if cfg == "convnext" else "TransformerDecoder"
        what = "ConvNextDenoiser (dim 512 x 20 blocks, mlp 4)" if cfg == "convnext" else "TransformerDecoderDenoiser (dim 512 x 12 layers, 8 heads, mlp 4)"
        w.metric = f"audio-seconds/sec/GPU ({n_steps}-step UniPC over the {fam} denoiser, mel only, 44.1 kHz / hop 512)"
        w.workload = (f"SURVEY 8(f) row 4: {what}"
                      f" behind the DENOISERS contract, {n_steps}-step UniPC, batch={B} x {seconds:g} s (T={T}), fresh features every step; features -> mel "
                      "(no vocoder pass)")
        w.cfg_extra = {"batch_per_gpu": B, "frames": T, "sampler": "unipc", "sampler_steps": n_steps}
        w.prof_handle = lambda: diff.denoise_fn.engine(dev)
        if cfg == "convnext":
            w.prof_kind, w.stride = _lib.PROF_CN_PWCONV1, args.prof_stride or 7
            w.kwhat = "pwconv1 (dim -> 4 dim) with the LayerNorm folded in and the GELU epilogue"
            w.traffic_key = "cn_pwconv1"
            w.alg_bytes = pwconv1_bytes(B * T, mc)
        else:
            w.prof_kind, w.stride = _lib.PROF_TD_ATTN, args.prof_stride or 7
            w.kwhat = "self- / cross-attention (QK^T + softmax + PV) of the decoder layers"
            w.traffic_key = "td_attn"
            w.alg_bytes = attention_bytes(T, B, mc)
        w.traffic_expect = {"config": cfg, "batch": B, "frames": T}
        w.diff = diff
    else:
        raise SystemExit(f"unknown config {cfg!r}")
    w.B, w.n_steps = B, n_steps
    return w
