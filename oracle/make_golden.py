"""Generate tests/golden/*.npz from the REAL reference (run in the build container only).

    python -m oracle.make_golden            # writes tests/golden/, asserts oracle == reference

TEST INFRASTRUCTURE ONLY.  Every fixture's *outputs* come from the unmodified reference classes
imported from /root/reference (oracle/_ref_import.py); the restatement in oracle/*_ref.py is
asserted equal to them on the same inputs (bit-exact on this machine).  The GPU box has no
/root/reference, so the parity tests there use (a) these committed vectors and (b) the pinned
restatement.  Big tensors (full-size weights, per-sample source noise) are not stored: they are
regenerated from the seeds recorded in the fixture and verified against the stored SHA-1.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import _ref_import, convnext_ref, features_ref, tfdec_ref, mel_ref, nsf_hifigan_ref, refinegan_ref, sampler_ref, wavenet_ref  # noqa: E402


def sha1_of(tensors) -> str:
    h = hashlib.sha1()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def state_sha1(sd) -> str:
    return sha1_of([sd[k] for k in sorted(sd)])


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(GOLD, name + ".npz")
    np.savez(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1e6:.2f} MB)")



# which make_golden section writes which fixture (prefix match), for tests/golden/MANIFEST.json
FIXTURE_SECTIONS = (("convnext_cross", "convnext_cross"), ("convnext", "convnext"), ("frontend_expand", "frontend_expand"), ("frontend_svs", "frontend_svs"),
                    ("tfdec", "tfdec"), ("refinegan_sine", "refinegan_sine"), ("nsf_v1_256_full", "round2"), ("chain_c1", "round2"), ("chain_c2", "round2"),
                    ("chain_", "round3"), ("ddpm1000_", "round3"), ("sampler_buffers", "round4"), ("svc_caller", "round4"),
                    ("nsf_rb2_", "round5"), ("mel_filterbank_hf", "round5"), ("sharded_c3_microbatch", "round6"))


def host_signature():
    """What fp32 CPU bits depend on besides the torch build: the CPU model (the BLAS picks its kernels by ISA) and the thread count.  Tests that
    assert BIT equality of an oracle against a fixture run only where this matches (tests/test_oracle_golden.py); everywhere else they hold
    the tolerance."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"cpu": model, "threads": torch.get_num_threads()}


def write_manifest():
    """tests/golden/MANIFEST.json: every fixture with its generating section, byte size, SHA-256 and array shapes -- checked by
    tests/test_oracle_golden.py::test_golden_manifest_lists_every_fixture, so a fixture cannot change or appear unrecorded."""
    import hashlib
    entries = {}
    for name in sorted(os.listdir(GOLD)):
        if not name.endswith(".npz"):
            continue
        path = os.path.join(GOLD, name)
        with open(path, "rb") as f:
            blob = f.read()
        with np.load(path, allow_pickle=False) as z:
            arrays = {k: [str(z[k].dtype), list(z[k].shape)] for k in z.files}
        section = next((sec for pre, sec in FIXTURE_SECTIONS if name.startswith(pre)), "main")
        entries[name] = {"section": f"python -m oracle.make_golden{'' if section == 'main' else ' ' + section}", "bytes": len(blob),
                         "sha256": hashlib.sha256(blob).hexdigest(), "arrays": arrays}
    manifest = {"generator": "oracle/make_golden.py (runs the REAL reference from /root/reference on seeded inputs; asserts oracle == reference)",
                "torch": torch.__version__, "numpy": np.__version__, "host": host_signature(), "fixtures": entries}
    with open(os.path.join(GOLD, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(f"  wrote MANIFEST.json ({len(entries)} fixtures)")


def synth_f0(T: int, frame_rate: float = 44100 / 512) -> torch.Tensor:
    """220*2^(0.3 sin(2 pi 0.7 t)) Hz with frames 100-130 unvoiced (scaled for short T)."""
    t = torch.arange(T, dtype=torch.float32) / frame_rate
    f0 = 220.0 * torch.pow(2.0, 0.3 * torch.sin(2 * np.pi * 0.7 * t))
    a, b = (100, 130) if T > 160 else (T // 3, T // 3 + max(2, T // 8))
    f0[a:b] = 0.0
    return f0


WN_SMALL = dict(mel_channels=128, d_encoder=256, residual_channels=64, residual_layers=4, dilation_cycle=4,
                use_linear_bias=True)
WN_FULL = dict(mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=20, dilation_cycle=4,
               use_linear_bias=True)  # configs/_base_/archs/diff_svc_v2.py:27-35


def build_ref_diffusion(R, wn_cfg, sd, **kw):
    diff = R["GaussianDiffusion"](denoiser=dict(type="WaveNetDenoiser", **wn_cfg), spec_min=[-5], spec_max=[0], **kw)
    diff.denoise_fn.load_state_dict(sd, strict=True)
    return diff.eval()


def oracle_denoiser(sd, cfg):
    return lambda x, t, c, xm, cm: wavenet_ref.wavenet_forward(
        sd, x, t, c, xm, cm, residual_layers=cfg["residual_layers"], dilation_cycle=cfg["dilation_cycle"])


CN_SMALL = dict(mel_channels=128, dim=64, mlp_factor=2, condition_dim=256, num_layers=4, dilation_cycle=4)
CN_FULL = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=20, dilation_cycle=4)   # convnext.py:156-163 defaults


@torch.no_grad()
def golden_convnext(R):
    """ConvNext denoiser: forward + the sampler loop driving it, outputs from the real reference classes."""
    print("convnext")

    def oracle_den(sd, cfg):
        return lambda x, t, c, xm, cm: convnext_ref.convnext_forward(sd, x, t, c, xm, cm, num_layers=cfg["num_layers"],
                                                                     dilation_cycle=cfg["dilation_cycle"])

    def shapes_kw(cfg):
        return {k: v for k, v in cfg.items() if k != "dilation_cycle"}

    for tag, cfg, seed, (B, T) in (("small", CN_SMALL, 301, (2, 50)), ("full", CN_FULL, 4321, (2, 96))):
        sd = convnext_ref.seeded_state(seed, **shapes_kw(cfg))
        net = R["ConvNext"](**cfg).eval()
        net.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, 128, T, generator=g)
        cond = torch.randn(B, 256, T, generator=g)
        t = torch.tensor([37.0, 912.5])[:B]
        masks = torch.zeros(B, T, dtype=torch.bool)
        masks[1, T - T // 4:] = True
        eps = net(x, t, cond)
        eps_masked = net(x, t, cond, x_masks=masks, cond_masks=masks)
        eps_long = net(x, torch.tensor([400], dtype=torch.long), cond)
        den = oracle_den(sd, cfg)
        assert torch.equal(den(x, t, cond, None, None), eps), "oracle ConvNext != reference"
        assert torch.equal(den(x, t, cond, masks, masks), eps_masked), "oracle ConvNext (masked) != reference"
        arrays = dict(x=x, cond=cond, t=t, masks=masks, eps=eps, eps_masked=eps_masked, eps_long=eps_long, seed=np.int64(seed),
                      weights_sha1=np.array(state_sha1(sd)))
        if tag == "small":
            arrays.update({"w:" + k: v for k, v in sd.items()})
        save(f"convnext_{tag}", **arrays)

    # the reference's own sampler loop driving the reference ConvNext (small net, every predictor, masks)
    sd = convnext_ref.seeded_state(301, **shapes_kw(CN_SMALL))
    diff = R["GaussianDiffusion"](denoiser=dict(type="ConvNextDenoiser", **CN_SMALL), spec_min=[-5], spec_max=[0]).eval()
    diff.denoise_fn.load_state_dict(sd, strict=True)
    den = oracle_den(sd, CN_SMALL)
    B, T = 2, 40
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(B, T, 256, generator=g)
    masks = torch.zeros(B, T, dtype=torch.bool)
    masks[1, 30:] = True
    for pred, interval in (("unipc", 50), ("plms", 50), ("naive", 100)):
        seed = 3000 + interval
        torch.manual_seed(seed)
        ref = diff(feats, sampler_interval=interval, noise_predictor=pred, x_masks=masks, cond_masks=masks)
        torch.manual_seed(seed)
        x_init = torch.randn(B, 128, T)
        n = len(range(0, 1000, interval))
        step_noise = torch.stack([torch.randn(B, 128, T) for _ in range(n)]) if pred == "naive" else torch.zeros(0)
        mine = sampler_ref.diffusion_sample(den, feats, x_init=x_init, sampler_interval=interval, predictor=pred,
                                            step_noise=step_noise, x_masks=masks, cond_masks=masks)
        assert torch.equal(mine, ref), f"oracle sampler over ConvNext {pred}/{interval} != reference"
        save(f"convnext_sampler_small_{pred}_i{interval}", features=feats, masks=masks, x_init=x_init, step_noise=step_noise, mel=ref,
             interval=np.int64(interval))

    # full-size net: 5 s, 20-step UniPC (the shape of BASELINE configs[0] with the denoiser swapped)
    sd = convnext_ref.seeded_state(4321, **shapes_kw(CN_FULL))
    diff = R["GaussianDiffusion"](denoiser=dict(type="ConvNextDenoiser", **CN_FULL), spec_min=[-5], spec_max=[0]).eval()
    diff.denoise_fn.load_state_dict(sd, strict=True)
    T, interval, seed = 430, 50, 4400
    feats = torch.randn(1, T, 256, generator=torch.Generator().manual_seed(seed))
    torch.manual_seed(seed + 100)
    ref = diff(feats, sampler_interval=interval)
    torch.manual_seed(seed + 100)
    x_init = torch.randn(1, 128, T)
    mine = sampler_ref.diffusion_sample(oracle_den(sd, CN_FULL), feats, x_init=x_init, sampler_interval=interval)
    assert torch.equal(mine, ref)
    save("convnext_sampler_full_c1", features=feats, x_init=x_init, mel=ref, interval=np.int64(interval), seed=np.int64(4321),
         weights_sha1=np.array(state_sha1(sd)))


@torch.no_grad()
def golden_frontend_expand(R):
    """The step before forward_features in SVCInference.forward: repeat_expand(text_features, mel_len).T and
    repeat_expand(pitches, mel_len) (tools/diffusion/inference.py:108-114), then forward_features -- reference functions."""
    print("front end: repeat_expand + forward_features")
    rexp = R["repeat_expand"]
    get_mask, fwd_features = R["diffsinger_methods"]()
    Enc = R["NaiveProjectionEncoder"]

    class RefFrontEnd(torch.nn.Module):
        forward_features = fwd_features

        def __init__(self, sd):
            super().__init__()
            self.get_mask_from_lengths = get_mask.__func__ if hasattr(get_mask, "__func__") else get_mask
            self.text_encoder = Enc(256, 256)
            self.speaker_encoder = Enc(10, 256, use_embedding=True)
            self.pitch_encoder = Enc(1, 256, preprocessing=R["pitch_to_scale"])
            self.load_state_dict(sd, strict=True)

    g = torch.Generator().manual_seed(77)
    arrays = {}
    # raw index semantics: up- and down-sampling, 1-D / 2-D
    for S, T in ((31, 70), (215, 430), (430, 215), (7, 7), (1, 9), (1000, 861)):
        x = torch.randn(3, S, generator=g)
        ref = rexp(x, T)                      # (the reference returns None for 3-D input: it falls off the end, tensor.py:38-43)
        assert torch.equal(features_ref.repeat_expand(x, T), ref)
        assert torch.equal(rexp(x[0], T), ref[0]) and torch.equal(features_ref.repeat_expand(x[None], T)[0], ref)
        arrays[f"x_{S}_{T}"], arrays[f"y_{S}_{T}"] = x, ref
    # the chain: extractor layout [Din, S] at its own frame rate, a pitch track of another length
    B, S, Sp, T = 2, 31, 50, 70
    contents_cf = torch.randn(B, 256, S, generator=g)
    f0_src = 80.0 + 600.0 * torch.rand(B, Sp, generator=g)
    ids = torch.tensor([4, 7])
    sd = features_ref.seeded_frontend_state(11)
    lens = torch.tensor([T, T])
    text = torch.stack([rexp(c, T).T for c in contents_cf])            # inference.py:113-114
    f0 = torch.stack([rexp(p, T) for p in f0_src])                      # :108-109
    ref = RefFrontEnd(sd).forward_features(ids, text, lens, T, mel_lens=lens, mel_max_len=T, pitches=f0.clone())
    mine = features_ref.forward_features(sd, torch.stack([features_ref.repeat_expand(c, T).T for c in contents_cf]), ids,
                                         torch.stack([features_ref.repeat_expand(p, T) for p in f0_src]), None, None, lens, T)
    assert torch.equal(mine["features"], ref["features"])
    arrays.update(contents_cf=contents_cf, f0_src=f0_src, ids=ids, features=ref["features"], T=np.int64(T),
                  sha1=np.array(state_sha1(sd)))
    save("frontend_expand", **arrays)


def golden_frontend_svs(R):
    """The SVS branch of the front end: `torch.gather(text_encoder(contents), 1, phones2mel) * (1 - mel_masks)`
    (archs/diffsinger/diffsinger.py:83-90; archs/hifisinger/core.py:71-79) and NaiveProjectionEncoder(use_neck=True)
    (modules/encoders/naive_projection.py:37-41) -- the reference's own method source on real encoder instances."""
    print("front end: phones2mel gather + use_neck encoders")
    get_mask, fwd_features = R["diffsinger_methods"]()
    Enc = R["NaiveProjectionEncoder"]
    Din, E, neck = 64, 96, 8

    class RefFrontEnd(torch.nn.Module):
        forward_features = fwd_features

        def __init__(self, sd, use_neck):
            super().__init__()
            self.get_mask_from_lengths = get_mask.__func__ if hasattr(get_mask, "__func__") else get_mask
            kw = dict(use_neck=True, neck_size=neck) if use_neck else {}
            self.text_encoder = Enc(Din, E, **kw)
            self.speaker_encoder = Enc(10, E, use_embedding=True)
            self.pitch_encoder = Enc(1, E, preprocessing=R["pitch_to_scale"], **kw)
            self.energy_encoder = Enc(1, E, **kw)
            self.load_state_dict(sd, strict=True)

    g = torch.Generator().manual_seed(123)
    B, S, T = 3, 23, 70                       # S phonemes, T mel frames
    contents = torch.randn(B, S, Din, generator=g)
    f0 = torch.rand(B, T, generator=g) * 1300.0
    energy = torch.rand(B, T, 1, generator=g)
    ids = torch.tensor([3, 0, 9])
    mel_lens = torch.tensor([70, 51, 64])
    src_lens = torch.tensor([23, 17, 20])
    phones2mel = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):                        # monotone durations like opencpop_transcription.py:50-57 (zeros in the padding)
        cuts = torch.sort(torch.randint(1, int(mel_lens[b]), (int(src_lens[b]) - 1,), generator=g)).values
        edges = [0] + cuts.tolist() + [int(mel_lens[b])]
        for i in range(int(src_lens[b])):
            phones2mel[b, edges[i]:edges[i + 1]] = i
    sd_neck = features_ref.seeded_svs_frontend_state(31, Din, E, 10, neck)
    sd_plain = features_ref.seeded_frontend_state(32, Din, E, 10, energy=True)
    arrays = dict(contents=contents, f0=f0, energy=energy, ids=ids, mel_lens=mel_lens, src_lens=src_lens, phones2mel=phones2mel,
                  neck=np.int64(neck), sha1_neck=np.array(state_sha1(sd_neck)), sha1_plain=np.array(state_sha1(sd_plain)))
    for tag, sd, use_neck, p2m in (("neck_gather", sd_neck, True, phones2mel), ("plain_gather", sd_plain, False, phones2mel),
                                   ("neck_frames", sd_neck, True, None)):
        c = contents if p2m is not None else torch.randn(B, T, Din, generator=g)
        if p2m is None:
            arrays["contents_frames"] = c
        ref = RefFrontEnd(sd, use_neck).forward_features(ids, c, src_lens if p2m is not None else mel_lens, S if p2m is not None else T,
                                                         mel_lens=mel_lens, mel_max_len=T, pitches=f0.clone(), phones2mel=p2m, energy=energy)
        mine = features_ref.forward_features(sd, c, ids, f0, None, energy, mel_lens, T, phones2mel=p2m)
        assert torch.equal(mine["features"], ref["features"]), tag
        arrays[f"features_{tag}"] = ref["features"]
    save("frontend_svs", **arrays)

    # HiFiSinger's copy of the gather (core.py:71-79): the mask is src_masks, taken over the mel frames
    get_mask_h, fwd_feat_h, _ = R["hifisinger_methods"]()
    hsd = features_ref.seeded_hifisinger_state(8, content_dim=Din, hidden=E)

    class RefHifi(torch.nn.Module):
        forward_features = fwd_feat_h

        def __init__(self):
            super().__init__()
            self.get_mask_from_lengths = get_mask_h.__func__ if hasattr(get_mask_h, "__func__") else get_mask_h
            self.text_encoder = Enc(Din, E)
            self.speaker_encoder = Enc(10, E, use_embedding=True)
            self.pitch_shift_encoder = Enc(1, E)
            self.energy_encoder = Enc(1, E)
            self.feature_fuser = torch.nn.Sequential(torch.nn.Linear(E, E), torch.nn.SiLU(), torch.nn.Linear(E, E), torch.nn.SiLU())
            self.load_state_dict(hsd, strict=True)

    shift = torch.randn(B, 1, generator=g)
    ref = RefHifi().forward_features(ids, contents, mel_lens, T, pitch_shift=shift, phones2mel=phones2mel, energy=energy)["features"]
    mine = features_ref.hifisinger_features(hsd, contents, ids, mel_lens, T, shift, energy, phones2mel=phones2mel)["features"]
    assert torch.equal(mine, ref), "oracle hifisinger gather != reference"
    save("frontend_svs_hifisinger", contents=contents, ids=ids, mel_lens=mel_lens, phones2mel=phones2mel, shift=shift, energy=energy,
         features=ref, sha1=np.array(state_sha1(hsd)))


TD_SMALL = dict(mel_channels=128, dim=128, mlp_factor=2, condition_dim=256, num_layers=2)
TD_FULL = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=12)   # convnext.py:264-271 defaults


@torch.no_grad()
def golden_tfdec(R):
    """TransformerDecoderDenoiser: forward + the sampler loop driving it, outputs from the real reference
    classes.  The restatement is asserted EQUAL since round 6: oracle/tfdec_ref.py::mha evaluates attention in the operation order torch's
    own MultiheadAttention takes (native fast path for self-attention, multi_head_attention_forward -> sdpa for cross-attention)."""
    print("transformer-decoder denoiser")
    close = torch.equal

    def oracle_den(sd, cfg):
        return lambda x, t, c, xm, cm: tfdec_ref.tfdec_forward(sd, x, t, c, xm, cm, num_layers=cfg["num_layers"])

    for tag, cfg, seed, (B, T) in (("small", TD_SMALL, 501, (2, 50)), ("full", TD_FULL, 5432, (2, 96))):
        sd = tfdec_ref.seeded_state(seed, **cfg)
        net = R["TransformerDecoderDenoiser"](**cfg).eval()
        net.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, 128, T, generator=g)
        cond = torch.randn(B, 256, T, generator=g)
        t = torch.tensor([37.0, 912.5])[:B]
        masks = torch.zeros(B, T, dtype=torch.bool)
        masks[1, T - T // 4:] = True
        eps = net(x, t, cond)
        eps_masked = net(x, t, cond, x_masks=masks, cond_masks=masks)
        eps_long = net(x, torch.tensor([400], dtype=torch.long), cond)
        den = oracle_den(sd, cfg)
        assert close(den(x, t, cond, None, None), eps), "oracle TransformerDecoderDenoiser != reference"
        assert close(den(x, t, cond, masks, masks), eps_masked), "oracle TransformerDecoderDenoiser (masked) != reference"
        assert close(den(x, torch.tensor([400], dtype=torch.long), cond, None, None), eps_long), "oracle TransformerDecoderDenoiser (long t) != reference"
        # (the sin/cos table is recomputed by whoever regenerates the weights: a different host CPU's vectorised sin / cos may
        # differ in the last bit, so it is left out of the fingerprint)
        arrays = dict(x=x, cond=cond, t=t, masks=masks, eps=eps, eps_masked=eps_masked, eps_long=eps_long, seed=np.int64(seed),
                      weights_sha1=np.array(state_sha1({k: v for k, v in sd.items() if k != "positional_embedding"})))
        save(f"tfdec_{tag}", **arrays)

    sd = tfdec_ref.seeded_state(501, **TD_SMALL)
    diff = R["GaussianDiffusion"](denoiser=dict(type="TransformerDecoderDenoiser", **TD_SMALL), spec_min=[-5], spec_max=[0]).eval()
    diff.denoise_fn.load_state_dict(sd, strict=True)
    den = oracle_den(sd, TD_SMALL)
    B, T = 2, 40
    g = torch.Generator().manual_seed(19)
    feats = torch.randn(B, T, 256, generator=g)
    masks = torch.zeros(B, T, dtype=torch.bool)
    masks[1, 30:] = True
    for pred, interval in (("unipc", 50), ("plms", 50), ("naive", 100)):
        seed = 5000 + interval
        torch.manual_seed(seed)
        ref = diff(feats, sampler_interval=interval, noise_predictor=pred, x_masks=masks, cond_masks=masks)
        torch.manual_seed(seed)
        x_init = torch.randn(B, 128, T)
        n = len(range(0, 1000, interval))
        step_noise = torch.stack([torch.randn(B, 128, T) for _ in range(n)]) if pred == "naive" else torch.zeros(0)
        mine = sampler_ref.diffusion_sample(den, feats, x_init=x_init, sampler_interval=interval, predictor=pred,
                                            step_noise=step_noise, x_masks=masks, cond_masks=masks)
        assert torch.equal(mine, ref), f"oracle sampler over tfdec {pred} != reference"
        save(f"tfdec_sampler_small_{pred}_i{interval}", features=feats, masks=masks, x_init=x_init, step_noise=step_noise, mel=ref,
             interval=np.int64(interval))


CNX_SMALL = dict(mel_channels=128, dim=128, mlp_factor=2, condition_dim=256, num_layers=6, dilation_cycle=4)     # cross blocks in front of layers 0 and 5
CNX_FULL = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=20, dilation_cycle=4)     # ... 0, 5, 10, 15


@torch.no_grad()
def golden_convnext_cross(R):
    """ConvNext(cross_attention=True) (convnext.py:95-152,186-193,246-250): forward (small / full-size, masked, long t) and the
    reference's sampler loop driving it (UniPC / PLMS with masks: PLMS's one unmasked call sees the unmasked condition as the
    attention memory).  Outputs from the real classes; the oracle restatement is asserted EQUAL (attention evaluated as torch does, see tfdec)."""
    print("convnext, cross-attention variant")
    EVERY = 5

    def oracle_den(sd, cfg):
        return lambda x, t, c, xm, cm: convnext_ref.convnext_forward(sd, x, t, c, xm, cm, num_layers=cfg["num_layers"],
                                                                     dilation_cycle=cfg["dilation_cycle"], cross_every=EVERY)

    def shapes_kw(cfg):
        return dict({k: v for k, v in cfg.items() if k != "dilation_cycle"}, cross_every=EVERY)

    close = torch.equal

    for tag, cfg, seed, (B, T) in (("small", CNX_SMALL, 311, (2, 50)), ("full", CNX_FULL, 4331, (2, 96))):
        sd = convnext_ref.seeded_state(seed, **shapes_kw(cfg))
        net = R["ConvNext"](cross_attention=True, cross_every_n_layers=EVERY, **cfg).eval()
        net.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, 128, T, generator=g)
        cond = torch.randn(B, 256, T, generator=g)
        t = torch.tensor([37.0, 912.5])[:B]
        masks = torch.zeros(B, T, dtype=torch.bool)
        masks[1, T - T // 4:] = True
        eps = net(x, t, cond)
        eps_masked = net(x, t, cond, x_masks=masks, cond_masks=masks)
        eps_long = net(x, torch.tensor([400], dtype=torch.long), cond)
        den = oracle_den(sd, cfg)
        assert close(den(x, t, cond, None, None), eps), "oracle ConvNext(cross) != reference"
        assert close(den(x, t, cond, masks, masks), eps_masked), "oracle ConvNext(cross, masked) != reference"
        # (the sin / cos table is recomputed on the test machine: libm may differ in the last bit across CPUs, so it stays out of the SHA)
        save(f"convnext_cross_{tag}", x=x, cond=cond, t=t, masks=masks, eps=eps, eps_masked=eps_masked, eps_long=eps_long, seed=np.int64(seed),
             weights_sha1=np.array(state_sha1({k: v for k, v in sd.items() if not k.endswith("positional_embedding")})))

    sd = convnext_ref.seeded_state(311, **shapes_kw(CNX_SMALL))
    diff = R["GaussianDiffusion"](denoiser=dict(type="ConvNextDenoiser", cross_attention=True, cross_every_n_layers=EVERY, **CNX_SMALL),
                                  spec_min=[-5], spec_max=[0]).eval()
    diff.denoise_fn.load_state_dict(sd, strict=True)
    den = oracle_den(sd, CNX_SMALL)
    B, T = 2, 40
    g = torch.Generator().manual_seed(19)
    feats = torch.randn(B, T, 256, generator=g)
    masks = torch.zeros(B, T, dtype=torch.bool)
    masks[1, 30:] = True
    for pred, interval in (("unipc", 50), ("plms", 50)):
        seed = 3100 + interval
        torch.manual_seed(seed)
        ref = diff(feats, sampler_interval=interval, noise_predictor=pred, x_masks=masks, cond_masks=masks)
        torch.manual_seed(seed)
        x_init = torch.randn(B, 128, T)
        mine = sampler_ref.diffusion_sample(den, feats, x_init=x_init, sampler_interval=interval, predictor=pred, x_masks=masks, cond_masks=masks)
        assert torch.equal(mine, ref), f"oracle sampler over ConvNext(cross) {pred}/{interval} != reference: {rel(mine, ref)}"
        save(f"convnext_cross_sampler_small_{pred}_i{interval}", features=feats, masks=masks, x_init=x_init, mel=ref, interval=np.int64(interval))


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@torch.no_grad()
def golden_refinegan_sine(R):
    """RefineGANGenerator(template_generator="sine") (generator.py:338-339; SineGen :197-310): outputs of the real module; the
    reference draws torch.rand(B, 1) (the zeroed initial phase, :254-257) before the noise tensors -- replayed in that order."""
    print("refinegan, sine template")
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fish_diffusion_refinegan_generator", os.path.join(_ref_import.REFERENCE_ROOT, "fish_diffusion/modules/vocoders/refinegan/generator.py"))
    rgmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rgmod)
    cfg = dict(refinegan_ref.CONFIG, template_generator="sine")
    for tag, seed, (B, T) in (("small", 23, (2, 5)), ("long", 24, (1, 173))):
        gen = rgmod.RefineGANGenerator(**cfg)
        gen.remove_weight_norm()
        gen.eval()
        rsd = refinegan_ref.seeded_state(seed, cfg)
        gen.load_state_dict(rsd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        mel = torch.randn(B, cfg["num_mels"], T, generator=g) * 0.5 - 2.0
        f0 = torch.stack([synth_f0(T, cfg["sampling_rate"] / cfg["hop_length"]) * (1 + 0.3 * b) for b in range(B)])[:, None]
        if tag == "long":
            f0[0, 0, 150:160] = 30000.0                  # above sr // 2: SineGen clears those samples (:277-278)
        torch.manual_seed(seed + 2)
        ref = gen(mel, f0)
        torch.manual_seed(seed + 2)
        torch.rand(B, 1)                                 # rand_ini (:254-256), zeroed for the only component (:257)
        shapes = refinegan_ref.noise_shapes(cfg, B, T)
        noises = [torch.randn((B, shapes[0][2], 1)).transpose(1, 2).contiguous()] + [torch.randn(sh) for sh in shapes[1:]]
        taps = {}
        mine = refinegan_ref.generator_forward(rsd, cfg, mel, f0, noises, taps)
        assert torch.equal(mine, ref), f"oracle refinegan sine {tag} != reference: {float((mine - ref).abs().max())}"
        save(f"refinegan_sine_{tag}", mel=mel, f0=f0, wav=ref, template=taps["template"], seed=np.int64(seed), noise_seed=np.int64(seed + 2),
             weights_sha1=np.array(state_sha1(rsd)), noise_sha1=np.array(sha1_of(noises)),
             config=np.array(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()})))


@torch.no_grad()
def golden_round2(R):
    """Round-2 fixtures.
    (a) nsf_v1_256_full: the vocoder layout configs/vocoder_nsf_hifigan.py:9,31 points at (tools/nsf_hifigan/config_v1_256.json)
        at FULL size -- 10 s, T = 1722 frames, 440 832 samples -- from the real `Generator` (models.py:353-448).
    (b) chain_c1 / chain_c2: the chain tools/diffusion/inference.py:140-160 runs -- features -> GaussianDiffusion (UniPC) -> mel
        -> `* 2.30259` (nsf_hifigan.py:79-80) -> Generator -> waveform -- on the real reference classes in fp32 (`wav`, `mel`), and
        the same chain with an fp64 DATA PATH (`wav64`, `mel64`: oracle restatement, weights / inputs / activations in float64, the
        schedule coefficients the reference's own fp32 values).  The fp64 run is the yardstick of the chained-parity test: an
        fp32 implementation cannot be asked to sit closer to another fp32 implementation than either sits to the exact result.
    """
    print("round 2: hop-256 full-size generator")
    h, seed, T = nsf_hifigan_ref.CONFIG_V1_256, 56, 1722
    gsd = nsf_hifigan_ref.seeded_generator_state(seed, h)
    gen = R["Generator"](R["AttrDict"](h))
    gen.remove_weight_norm()
    gen.eval()
    gen.load_state_dict(gsd, strict=True)
    g = torch.Generator().manual_seed(seed + 1)
    mel = torch.randn(1, 128, T, generator=g) * 0.5 - 2.0
    f0 = synth_f0(T, h["sampling_rate"] / h["hop_size"])[None]
    L = T * h["hop_size"]
    torch.manual_seed(seed + 2)
    ref = gen(mel, f0)
    torch.manual_seed(seed + 2)
    rand_ini = torch.rand(1, 9)
    rand_ini[:, 0] = 0
    src_noise = torch.randn(1, L, 9)
    mine = nsf_hifigan_ref.generator_forward(gsd, h, mel, f0, rand_ini, src_noise)
    assert torch.equal(mine, ref), "oracle generator v1_256_full != reference"
    save("nsf_v1_256_full", mel=mel, f0=f0, rand_ini=rand_ini, wav=ref, seed=np.int64(seed), noise_seed=np.int64(seed + 2),
         weights_sha1=np.array(state_sha1(gsd)), src_noise_sha1=np.array(sha1_of([src_noise])), config=np.array(json.dumps(h)))

    print("round 2: chained features -> waveform (fp32 reference + fp64 data path)")
    sd = wavenet_ref.seeded_wavenet_state(1234, **{k: v for k, v in WN_FULL.items() if k != "dilation_cycle"})
    diff = build_ref_diffusion(R, WN_FULL, sd)
    hv = nsf_hifigan_ref.CONFIG_V1
    vsd = nsf_hifigan_ref.seeded_generator_state(55, hv)
    gen = R["Generator"](R["AttrDict"](hv))
    gen.remove_weight_norm()
    gen.eval()
    gen.load_state_dict(vsd, strict=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    vsd64 = {k: v.double() for k, v in vsd.items()}
    den64_ = oracle_denoiser(sd64, WN_FULL)
    den64 = lambda x, t, c, xm, cm: den64_(x, t.double(), c, xm, cm)   # noqa: E731  (step embedding in fp64 as well)
    for tag, T, interval, seed in (("c1", 430, 50, 1234), ("c2", 861, 10, 1235)):
        gg = torch.Generator().manual_seed(seed)
        feats = torch.randn(1, T, 256, generator=gg)
        f0 = synth_f0(T)[None]
        torch.manual_seed(seed + 100)
        mel_ref = diff(feats, sampler_interval=interval)                      # [1, T, 128], log10 scale (diff_svc_v2)
        torch.manual_seed(seed + 200)
        wav_ref = gen(2.30259 * mel_ref.transpose(1, 2), f0)                  # spec2wav: use_natural_log=False rescale
        torch.manual_seed(seed + 100)
        x_init = torch.randn(1, 128, T)
        torch.manual_seed(seed + 200)
        rand_ini = torch.rand(1, 9)
        rand_ini[:, 0] = 0
        src_noise = torch.randn(1, T * hv["hop_size"], 9)
        mel64 = sampler_ref.diffusion_sample(den64, feats.double(), x_init=x_init.double(), sampler_interval=interval)
        assert mel64.dtype == torch.float64
        wav64 = nsf_hifigan_ref.generator_forward(vsd64, hv, 2.30259 * mel64.transpose(1, 2), f0.double(), rand_ini.double(),
                                                  src_noise.double())
        assert wav64.dtype == torch.float64
        # the vocoder alone on the REFERENCE's fp32 mel, fp64 data path: separates "mel noise amplified by the vocoder" from vocoder noise
        wav64_of_ref_mel = nsf_hifigan_ref.generator_forward(vsd64, hv, (2.30259 * mel_ref.transpose(1, 2)).double(), f0.double(),
                                                             rand_ini.double(), src_noise.double())
        e_mel = float((mel_ref.double() - mel64).abs().max() / mel64.abs().max())
        e_wav = float((wav_ref.double() - wav64).abs().max())
        print(f"  chain_{tag}: reference fp32 vs fp64 data path: mel rel {e_mel:.3e}, wav abs {e_wav:.3e} "
              f"(vocoder alone on the same mel: {float((wav_ref.double() - wav64_of_ref_mel).abs().max()):.3e})")
        save(f"chain_{tag}", features=feats, f0=f0, x_init=x_init, rand_ini=rand_ini, mel=mel_ref, wav=wav_ref,
             mel64=mel64.float(), wav64=wav64.float(),   # the fp64 results, STORED as fp32 (6e-8 of storage rounding against errors of 1e-5..1e-4)
             interval=np.int64(interval), noise_seed=np.int64(seed + 200),
             src_noise_sha1=np.array(sha1_of([src_noise])), wn_sha1=np.array(state_sha1(sd)), voc_sha1=np.array(state_sha1(vsd)),
             ref_vs_f64_wav_abs=np.float64(e_wav), ref_vs_f64_mel_rel=np.float64(e_mel))


def _ref_frontend(R, sd, n_speakers=10):
    """A module carrying REAL reference encoder instances whose `forward_features` / `get_mask_from_lengths` are the reference's own
    method source (archs/diffsinger/diffsinger.py:41-134) -- DiffSinger.__init__ itself pulls lightning / wandb and is skipped."""
    get_mask, fwd_features = R["diffsinger_methods"]()
    Enc = R["NaiveProjectionEncoder"]

    class RefFrontEnd(torch.nn.Module):
        forward_features = fwd_features

        def __init__(self):
            super().__init__()
            self.get_mask_from_lengths = get_mask.__func__ if hasattr(get_mask, "__func__") else get_mask
            self.text_encoder = Enc(256, 256)
            self.speaker_encoder = Enc(n_speakers, 256, use_embedding=True)
            self.pitch_encoder = Enc(1, 256, preprocessing=R["pitch_to_scale"])
            self.load_state_dict(sd, strict=True)

    return RefFrontEnd().eval()


@torch.no_grad()
def golden_round3(R):
    """Round-3 fixtures (VERDICT r2 "missing 2", "weak 3").
    (a) ddpm1000_full_T430 / ddpm1000_full_T861: BASELINE configs[4]'s sampler AT FULL SIZE from the real `GaussianDiffusion`:
        `noise_predictor="naive"`, `sampler_interval=1` => 1000 denoiser calls (diffusions/diffusion.py:246-253,
        noise_predictor.py:73-104), diff_svc_v2 WaveNet C = 512 x 20 layers, batch 1, 5 s and 10 s.  The 1000 injected noises
        (220 / 441 MB) are not stored: `ddpm_noise(seed, ...)` regenerates the reference's own draw sequence, verified by SHA-1.
    (b) ddpm1000_spk_chain: the multi-speaker shape of configs[4] -- the reference's `DiffSinger.forward_features` (speaker
        embedding + content + pitch encoders, its own masks for two different lengths) feeding the same 1000-step run, batch 2.
    (c) chain_c3 / c4 / c5: three more seeded draws of the chained features -> 100-step UniPC -> NSF-HiFiGAN -> waveform fixture
        (golden_round2 (b)), so that the 1e-4 chained bar is shown on five draws instead of two.
    """
    import time
    sd = wavenet_ref.seeded_wavenet_state(1234, **{k: v for k, v in WN_FULL.items() if k != "dilation_cycle"})
    diff = build_ref_diffusion(R, WN_FULL, sd)
    den = oracle_denoiser(sd, WN_FULL)
    for T, seed, check_oracle in ((430, 4301, True), (861, 4302, False)):
        print(f"round 3: full-size 1000-step DDPM, T = {T}")
        feats = torch.randn(1, T, 256, generator=torch.Generator().manual_seed(seed))
        t0 = time.perf_counter()
        torch.manual_seed(seed + 100)
        ref = diff(feats, sampler_interval=1, noise_predictor="naive")
        print(f"  reference: {time.perf_counter() - t0:.1f} s")
        x_init, step_noise = sampler_ref.ddpm_noise(seed + 100, 1, 128, T, 1000)
        if check_oracle:
            mine = sampler_ref.diffusion_sample(den, feats, x_init=x_init, sampler_interval=1, predictor="naive", step_noise=step_noise)
            assert torch.equal(mine, ref), "oracle 1000-step DDPM (full net) != reference"
        # T = 861: the oracle's equality with the reference is the T = 430 case (same code, same draw order); here the reference runs alone
        save(f"ddpm1000_full_T{T}", features=feats, x_init=x_init, mel=ref, interval=np.int64(1), noise_seed=np.int64(seed + 100),
             step_noise_sha1=np.array(sha1_of([step_noise])), step_noise_first=step_noise[0, 0, :4, :8].clone(),
             weights_seed=np.int64(1234), weights_sha1=np.array(state_sha1(sd)))
        del step_noise

    print("round 3: multi-speaker front end -> 1000-step DDPM (full net), batch 2 with masks")
    sd_f = features_ref.seeded_frontend_state(11)
    fe = _ref_frontend(R, sd_f)
    g = torch.Generator().manual_seed(4310)
    B, T = 2, 215
    contents = torch.randn(B, T, 256, generator=g)
    f0 = torch.stack([synth_f0(T), synth_f0(T) * 1.5])
    lens = torch.tensor([215, 176])
    spk = torch.tensor([7, 2])
    fr = fe.forward_features(spk, contents, lens, T, mel_lens=lens, mel_max_len=T, pitches=f0.clone())
    mine_f = features_ref.forward_features(sd_f, contents, spk, f0, None, None, lens, T)
    assert torch.equal(mine_f["features"], fr["features"]) and torch.equal(mine_f["x_masks"], fr["x_masks"])
    torch.manual_seed(4311)
    ref = diff(fr["features"], sampler_interval=1, noise_predictor="naive", x_masks=fr["x_masks"], cond_masks=fr["cond_masks"])
    x_init, step_noise = sampler_ref.ddpm_noise(4311, B, 128, T, 1000)
    mine = sampler_ref.diffusion_sample(den, fr["features"], x_init=x_init, sampler_interval=1, predictor="naive", step_noise=step_noise,
                                        x_masks=fr["x_masks"], cond_masks=fr["cond_masks"])
    assert torch.equal(mine, ref), "oracle multi-speaker 1000-step DDPM != reference"
    save("ddpm1000_spk_chain", contents=contents, f0=f0, lens=lens, speakers=spk, features=fr["features"], masks=fr["x_masks"],
         x_init=x_init, mel=ref, noise_seed=np.int64(4311), step_noise_sha1=np.array(sha1_of([step_noise])),
         frontend_sha1=np.array(state_sha1(sd_f)), weights_sha1=np.array(state_sha1(sd)))
    del step_noise

    print("round 3: three more chained features -> waveform draws")
    hv = nsf_hifigan_ref.CONFIG_V1
    vsd = nsf_hifigan_ref.seeded_generator_state(55, hv)
    gen = R["Generator"](R["AttrDict"](hv))
    gen.remove_weight_norm()
    gen.eval()
    gen.load_state_dict(vsd, strict=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    vsd64 = {k: v.double() for k, v in vsd.items()}
    den64_ = oracle_denoiser(sd64, WN_FULL)
    den64 = lambda x, t, c, xm, cm: den64_(x, t.double(), c, xm, cm)   # noqa: E731
    for tag, T, interval, seed in (("c3", 861, 10, 1236), ("c4", 645, 10, 1237), ("c5", 517, 10, 1238)):
        gg = torch.Generator().manual_seed(seed)
        feats = torch.randn(1, T, 256, generator=gg)
        f0 = synth_f0(T)[None] * (1.0 + 0.25 * (seed - 1236))
        torch.manual_seed(seed + 100)
        mel_ref = diff(feats, sampler_interval=interval)
        torch.manual_seed(seed + 200)
        wav_ref = gen(2.30259 * mel_ref.transpose(1, 2), f0)
        torch.manual_seed(seed + 100)
        x_init = torch.randn(1, 128, T)
        torch.manual_seed(seed + 200)
        rand_ini = torch.rand(1, 9)
        rand_ini[:, 0] = 0
        src_noise = torch.randn(1, T * hv["hop_size"], 9)
        mel64 = sampler_ref.diffusion_sample(den64, feats.double(), x_init=x_init.double(), sampler_interval=interval)
        wav64 = nsf_hifigan_ref.generator_forward(vsd64, hv, 2.30259 * mel64.transpose(1, 2), f0.double(), rand_ini.double(),
                                                  src_noise.double())
        e_mel = float((mel_ref.double() - mel64).abs().max() / mel64.abs().max())
        e_wav = float((wav_ref.double() - wav64).abs().max())
        print(f"  chain_{tag}: reference fp32 vs fp64 data path: mel rel {e_mel:.3e}, wav abs {e_wav:.3e}")
        save(f"chain_{tag}", features=feats, f0=f0, x_init=x_init, rand_ini=rand_ini, mel=mel_ref, wav=wav_ref,
             mel64=mel64.float(), wav64=wav64.float(), interval=np.int64(interval), noise_seed=np.int64(seed + 200),
             src_noise_sha1=np.array(sha1_of([src_noise])), wn_sha1=np.array(state_sha1(sd)), voc_sha1=np.array(state_sha1(vsd)),
             ref_vs_f64_wav_abs=np.float64(e_wav), ref_vs_f64_mel_rel=np.float64(e_mel))


@torch.no_grad()
def golden_round5(R):
    """Round-5 fixtures (VERDICT r4 missing 2 / 5).
    (a) nsf_rb2_small / nsf_rb2_long: the real `Generator` built with `resblock="2"` (models.py:119-158: two-conv ResBlock2, one conv per
        dilation, `xt + x`), the one product branch whose oracle restatement (nsf_hifigan_ref.py `_resblock2`) had never met the reference.
        Two layouts: config_v1 with [[1, 3]] x 3 (the usual ResBlock2 shape) batched with an unvoiced item, and a longer single item
        with uneven kernel sizes / dilations (the widest reach, 5 x 5 = 25 columns, inside the library's 32-column halo).
    (b) mel_filterbank_hf: the slaney-normalised mel filterbank from an implementation that is NOT ours -- `transformers.audio_utils.
        mel_filter_bank(norm="slaney", mel_scale="slaney")`, HuggingFace's numpy restatement of `librosa.filters.mel` which its own test
        suite holds against librosa -- for the four (sr, n_fft) geometries the key-shift path visits (pitch_adjustable_mel.py:34-53).  librosa
        itself is not installable here; this is the independent table the product's `fdx_mel_filterbank` and `oracle/mel_ref.py` are held to.
    """
    print("round 5: Generator(resblock='2') from the real reference")
    for tag, h, seed, (B, T) in (
            ("small", dict(nsf_hifigan_ref.CONFIG_V1, resblock="2", resblock_dilation_sizes=[[1, 3], [1, 3], [1, 3]]), 91, (3, 11)),
            ("long", dict(nsf_hifigan_ref.CONFIG_V1, resblock="2", resblock_kernel_sizes=[3, 5, 11], resblock_dilation_sizes=[[1, 2], [2, 6], [3, 5]]), 92, (1, 173))):
        gsd = nsf_hifigan_ref.seeded_generator_state(seed, h)
        gen = R["Generator"](R["AttrDict"](h))
        assert type(gen.resblocks[0]).__name__ == "ResBlock2"
        gen.remove_weight_norm()
        gen.eval()
        gen.load_state_dict(gsd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        mel = torch.randn(B, 128, T, generator=g) * 0.5 - 2.0
        f0 = torch.stack([synth_f0(T, h["sampling_rate"] / h["hop_size"]) * (1 + 0.5 * i) for i in range(B)])
        if B > 2:
            f0[2] = 0.0
        torch.manual_seed(seed + 2)
        ref = gen(mel, f0)
        torch.manual_seed(seed + 2)
        rand_ini = torch.rand(B, 9)
        rand_ini[:, 0] = 0
        src_noise = torch.randn(B, T * h["hop_size"], 9)
        mine = nsf_hifigan_ref.generator_forward(gsd, h, mel, f0, rand_ini, src_noise)
        assert torch.equal(mine, ref), f"oracle generator resblock2 {tag} != reference"
        save(f"nsf_rb2_{tag}", mel=mel, f0=f0, rand_ini=rand_ini, wav=ref, seed=np.int64(seed), noise_seed=np.int64(seed + 2),
             weights_sha1=np.array(state_sha1(gsd)), src_noise_sha1=np.array(sha1_of([src_noise])), config=np.array(json.dumps(h)))

    print("round 5: third-party slaney filterbank (transformers.audio_utils.mel_filter_bank)")
    # in a clean interpreter: this process carries _ref_import's librosa STUB in sys.modules, which transformers would mistake for the real package
    import subprocess
    import tempfile
    geoms = [mel_ref.stft_geometry(2048, 2048, 512, ks, 1.0)[0] for ks in (0, 3, -5, 12)]
    with tempfile.TemporaryDirectory() as td:
        code = ("import sys, numpy as np, transformers\nfrom transformers.audio_utils import mel_filter_bank\n"
                "out = {str(n): mel_filter_bank(num_frequency_bins=n // 2 + 1, num_mel_filters=128, min_frequency=40, max_frequency=16000, "
                "sampling_rate=44100, norm='slaney', mel_scale='slaney').T for n in map(int, sys.argv[2:])}\n"
                "np.savez(sys.argv[1], version=np.array(transformers.__version__), **out)\n")
        subprocess.run([sys.executable, "-c", code, os.path.join(td, "hf.npz"), *map(str, geoms)], check=True)
        hf = dict(np.load(os.path.join(td, "hf.npz")))
    arrays = {}
    for n_fft_new in geoms:
        fb = hf[str(n_fft_new)]                                                                  # [128, bins] float64
        mine = mel_ref.slaney_mel_filterbank(sr=44100, n_fft=n_fft_new, n_mels=128, fmin=40, fmax=16000)
        err = float(np.abs(fb - mine).max() / np.abs(fb).max())
        print(f"  n_fft {n_fft_new}: oracle vs transformers {err:.2e} of the peak")
        assert err < 5e-7
        arrays[f"fb_nfft{n_fft_new}"] = fb.astype(np.float32)
    save("mel_filterbank_hf", n_ffts=np.array(geoms, dtype=np.int64), source=np.array(f"transformers {hf['version']} audio_utils.mel_filter_bank"), **arrays)


@torch.no_grad()
def golden_round6(R):
    """Round 6: BASELINE configs[3] at bench scale IN ONE PIECE.  `sharded_c3_microbatch`: rank 0's first micro-batch of the sharded bench job
    (bench.py --config sharded: 64 seeded lengths in [516, 861] dealt longest-first over 8 ranks, micro-batches of <= 8) -- 8 utterances
    through the REAL reference, one at a time (an exact-ragged micro-batch is defined as every utterance's batch-1 result): features ->
    GaussianDiffusion (full 20-layer WaveNet, 100-step UniPC) -> mel for each of the 8; and for three of them (shortest, median, longest)
    mel -> `* 2.30259` -> NSF-HiFiGAN Generator -> waveform.  Inputs are regenerated from the stored seeds (SHA-1 checked)."""
    print("round 6: configs[3] micro-batch, full size (8 reference runs of 100 UniPC steps: minutes)")
    all_lens = torch.randint(516, 862, (64,), generator=torch.Generator().manual_seed(4)).tolist()      # benchkit/workloads.py, `sharded`
    order = sorted(range(64), key=lambda i: (-all_lens[i], i))                                         # dist.shard_utterances: longest first, dealt round-robin
    mine = order[0::8]
    lens = [all_lens[i] for i in mine]                                                                 # rank 0's 8 utterances = its one micro-batch
    sd = wavenet_ref.seeded_wavenet_state(1234, **{k: v for k, v in WN_FULL.items() if k != "dilation_cycle"})
    diff = build_ref_diffusion(R, WN_FULL, sd)
    hv = nsf_hifigan_ref.CONFIG_V1
    vsd = nsf_hifigan_ref.seeded_generator_state(55, hv)
    gen = R["Generator"](R["AttrDict"](hv))
    gen.remove_weight_norm()
    gen.eval()
    gen.load_state_dict(vsd, strict=True)
    by_len = sorted(range(8), key=lambda b: lens[b])
    voc_items = [by_len[0], by_len[4], by_len[7]]
    arrays = dict(lens=np.array(lens, np.int64), utterance_ids=np.array(mine, np.int64), interval=np.int64(10), voc_items=np.array(voc_items, np.int64),
                  wn_sha1=np.array(state_sha1(sd)), voc_sha1=np.array(state_sha1(vsd)))
    feat_sha, noise_sha = [], []
    for b, n in enumerate(lens):
        feats = torch.randn(1, n, 256, generator=torch.Generator().manual_seed(6000 + b))
        torch.manual_seed(6100 + b)
        mel_ref = diff(feats, sampler_interval=10)                            # [1, n, 128]
        arrays[f"mel_{b}"] = mel_ref[0]
        feat_sha.append(sha1_of([feats]))
        print(f"  utterance {b}: {n} frames, mel range [{float(mel_ref.min()):.2f}, {float(mel_ref.max()):.2f}]")
        if b in voc_items:
            f0 = synth_f0(n)[None]
            torch.manual_seed(6200 + b)
            wav = gen(2.30259 * mel_ref.transpose(1, 2), f0)
            torch.manual_seed(6200 + b)
            rand_ini = torch.rand(1, 9)
            rand_ini[:, 0] = 0
            src_noise = torch.randn(1, n * hv["hop_size"], 9)
            assert torch.equal(nsf_hifigan_ref.generator_forward(vsd, hv, 2.30259 * mel_ref.transpose(1, 2), f0, rand_ini, src_noise), wav)
            arrays[f"wav_{b}"] = wav[0, 0]
            arrays[f"rand_ini_{b}"] = rand_ini
            noise_sha.append(sha1_of([src_noise]))
    arrays["features_sha1"] = np.array(feat_sha)
    arrays["src_noise_sha1"] = np.array(noise_sha)
    save("sharded_c3_microbatch", **arrays)


def golden_round4(R):
    """Round-4 fixtures.
    (a) sampler_buffers: in the reference the DDPM / PLMS coefficients are `register_buffer`s of the predictor modules
        (noise_predictor.py:29-71,115) -- what a checkpoint holds is what sampling uses.  A real `GaussianDiffusion` whose predictor
        buffers were overwritten (as `load_state_dict` would) runs the naive and PLMS samplers; the fixture holds the buffers and the
        reference's mels.  The oracle takes the same buffers (`naive_buffers=`, `plms_alphas_cumprod=`) and must equal the reference.
    (b) svc_caller: the reference's own caller, `SVCInference.forward` (tools/diffusion/inference.py:86-162), run HERE over the reference's
        own modules with stub extractors: reference front end (`DiffSinger.forward_features`), `GaussianDiffusion` + `WaveNet`, the NSF-HiFiGAN
        `Generator` behind `spec2wav`'s three scalar lines (nsf_hifigan.py:72-85), an `ema_model` whose weights differ from `model`'s.  The
        fixture holds the stub extractors' outputs and the waveform; `tests/golden/svc_inference_forward.json` holds the SOURCE TEXT of
        that one method (with the file's SHA-256 and line range) so that the GPU box -- which has no reference tree -- can run the same
        body over the installed MI355X modules.  Test infrastructure only: nothing in the product reads it.
    """
    import ast
    import hashlib as _hl
    from typing import Optional as _Optional
    print("round 4: sampler coefficient buffers as a checkpoint holds them")
    cfg = WN_SMALL
    sd = wavenet_ref.seeded_wavenet_state(77, **{k: v for k, v in cfg.items() if k != "dilation_cycle"})
    diff = build_ref_diffusion(R, cfg, sd)
    nb = diff.naive_noise_predictor
    nb.posterior_mean_coef1.mul_(1.03)
    nb.sqrt_recipm1_alphas_cumprod.mul_(0.98)
    nb.posterior_log_variance_clipped.add_(0.2)
    nb.clip_min.fill_(-0.9)
    nb.clip_max.fill_(0.8)
    diff.plms_noise_predictor.alphas_cumprod.pow_(1.05)
    den = oracle_denoiser(sd, cfg)
    B, T, interval = 2, 60, 50
    feats = torch.randn(B, T, 256, generator=torch.Generator().manual_seed(771))
    torch.manual_seed(772)
    mel_naive = diff(feats, sampler_interval=interval, noise_predictor="naive")
    x_naive, step_noise = sampler_ref.ddpm_noise(772, B, 128, T, 1000 // interval)
    nbuf = {k: v.clone() for k, v in nb.state_dict().items()}
    mine = sampler_ref.diffusion_sample(den, feats, x_init=x_naive, sampler_interval=interval, predictor="naive", step_noise=step_noise, naive_buffers=nbuf)
    assert torch.equal(mine, mel_naive), "oracle naive sampler with loaded buffers != reference"
    plain = sampler_ref.diffusion_sample(den, feats, x_init=x_naive, sampler_interval=interval, predictor="naive", step_noise=step_noise)
    assert not torch.equal(plain, mel_naive), "the perturbed buffers must change the result"
    torch.manual_seed(773)
    mel_plms = diff(feats, sampler_interval=interval, noise_predictor="plms")
    torch.manual_seed(773)
    x_plms = torch.randn(B, 128, T)
    acp = diff.plms_noise_predictor.alphas_cumprod.clone()
    mine = sampler_ref.diffusion_sample(den, feats, x_init=x_plms, sampler_interval=interval, predictor="plms", plms_alphas_cumprod=acp)
    assert torch.equal(mine, mel_plms), "oracle PLMS sampler with a loaded alphas_cumprod != reference"
    print(f"  perturbed vs schedule-derived coefficients: naive mel differs by {float((plain - mel_naive).abs().max()):.3f}")
    save("sampler_buffers", features=feats, x_naive=x_naive, step_noise=step_noise, x_plms=x_plms, mel_naive=mel_naive, mel_plms=mel_plms,
         interval=np.int64(interval), weights_seed=np.int64(77), weights_sha1=np.array(state_sha1(sd)), plms_alphas_cumprod=acp,
         **{"naive:" + k: v for k, v in nbuf.items()})

    print("round 4: SVCInference.forward (the reference's caller) over the reference's modules")
    path = os.path.join(_ref_import.REFERENCE_ROOT, "tools/diffusion/inference.py")
    with open(path) as f:
        text = f.read()
    tree = ast.parse(text)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SVCInference")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward")
    first = min([fn.lineno] + [d.lineno for d in fn.decorator_list])
    lines = text.splitlines()[first - 1:fn.end_lineno]
    indent = len(lines[0]) - len(lines[0].lstrip())
    source = "\n".join(ln[indent:] for ln in lines) + "\n"
    with open(os.path.join(GOLD, "svc_inference_forward.json"), "w") as f:
        json.dump({"what": "source text of ONE method of the reference, `SVCInference.forward`, extracted by oracle/make_golden.py round4 so that the GPU box "
                           "(no reference tree) can run the reference's own caller body over the installed MI355X modules; test infrastructure only",
                   "reference_file": "tools/diffusion/inference.py", "lines": [first, fn.end_lineno],
                   "file_sha256": _hl.sha256(text.encode()).hexdigest(), "source": source}, f, indent=1)
    ns = {"torch": torch, "np": np, "Optional": _Optional, "repeat_expand": R["repeat_expand"]}
    exec(compile(source, path, "exec"), ns)
    fwd = ns["forward"]

    wcfg = WN_SMALL
    fe_model, fe_ema = features_ref.seeded_frontend_state(81), features_ref.seeded_frontend_state(82)
    wn_model = wavenet_ref.seeded_wavenet_state(83, **{k: v for k, v in wcfg.items() if k != "dilation_cycle"})
    wn_ema = wavenet_ref.seeded_wavenet_state(84, **{k: v for k, v in wcfg.items() if k != "dilation_cycle"})
    hv = nsf_hifigan_ref.CONFIG_V1
    vsd = nsf_hifigan_ref.seeded_generator_state(85, hv)
    gen = R["Generator"](R["AttrDict"](hv))
    gen.remove_weight_norm()
    gen.eval()
    gen.load_state_dict(vsd, strict=True)

    def ref_diffsinger(fe_sd, wn_sd):
        m = _ref_frontend(R, fe_sd)
        m.diffusion = build_ref_diffusion(R, wcfg, wn_sd)
        return m.eval()

    class RefVocoder:                       # nsf_hifigan.py:72-85 around the real Generator (the wrapper class itself imports lightning)
        use_natural_log = False

        def spec2wav(self, mel, f0, key_shift=0):
            c = mel[None]
            if key_shift is not None and key_shift != 0:
                f0 *= 2 ** (key_shift / 12)
            if self.use_natural_log is False:
                c = 2.30259 * c
            f0 = f0[None].to(c.dtype)
            return gen(c, f0).view(-1)

    class Lightning:                        # what load_checkpoint returns there: .model, .ema_model, .vocoder
        pass

    class ModelCfg(dict):
        pass

    class Holder(torch.nn.Module):          # the attributes SVCInference.forward touches
        forward = fwd

        def __init__(self, text_features, pitches):
            super().__init__()
            self.anchor = torch.nn.Parameter(torch.zeros(1))      # `self.device` = next(self.parameters()).device
            self.config = type("Cfg", (), {"model": ModelCfg()})()
            self.model = Lightning()
            self.model.model = ref_diffsinger(fe_model, wn_model)
            self.model.ema_model = ref_diffsinger(fe_ema, wn_ema)
            self.model.vocoder = RefVocoder()
            self.text_features_extractor = lambda audio, sr: text_features.clone()
            self.pitch_extractor = lambda audio, sr, pad_to=None: pitches[:pad_to].clone()

        @property
        def device(self):
            return next(self.parameters()).device

    T, S = 70, 112
    g = torch.Generator().manual_seed(86)
    text_features = torch.randn(1, 256, S, generator=g)           # an extractor at ~50 frames/s; forward() nearest-expands it to mel_len
    pitches = synth_f0(T)
    audio = torch.zeros(1, T * 512 + 100)                          # only its length is read by the caller (mel_len = n // 512)
    holder = Holder(text_features, pitches).eval()
    torch.manual_seed(87)
    wav = holder(audio, 44100, pitch_adjust=2, speakers=torch.tensor([3]), sampler_interval=50)
    wav = torch.from_numpy(np.asarray(wav))
    torch.manual_seed(87)
    wav_again = torch.from_numpy(np.asarray(holder(audio, 44100, pitch_adjust=2, speakers=torch.tensor([3]), sampler_interval=50)))
    assert torch.equal(wav, wav_again) and wav.shape == (T * 512,)
    # the same chain through the oracle: front end (EMA weights) -> 20-step UniPC -> vocoder, draws regenerated in the reference's order
    torch.manual_seed(87)
    x_T = torch.randn(1, 128, T)
    rand_ini = torch.rand(1, 9)
    rand_ini[:, 0] = 0
    src_noise = torch.randn(1, T * 512, 9)
    f0 = pitches * 2 ** (2 / 12)
    contents = features_ref.repeat_expand(text_features[0], T).T[None]
    feats = features_ref.forward_features(fe_ema, contents, speakers=torch.tensor([3]), pitches=f0[None], mel_lens=torch.tensor([T]))["features"]
    mel = sampler_ref.diffusion_sample(oracle_denoiser(wn_ema, wcfg), feats, x_init=x_T, sampler_interval=50)
    mine = nsf_hifigan_ref.generator_forward(vsd, hv, 2.30259 * mel[0].T[None], f0[None], rand_ini, src_noise).view(-1)
    assert torch.equal(mine, wav), f"oracle chain != SVCInference.forward over the reference modules ({float((mine - wav).abs().max()):.3e})"
    save("svc_caller", text_features=text_features, pitches=pitches, n_audio=np.int64(audio.shape[-1]), speaker=np.int64(3), pitch_adjust=np.int64(2),
         sampler_interval=np.int64(50), noise_seed=np.int64(87), wav=wav, mel=mel,
         seeds=np.array([81, 82, 83, 84, 85], dtype=np.int64),
         sha1=np.array([state_sha1(fe_model), state_sha1(fe_ema), state_sha1(wn_model), state_sha1(wn_ema), state_sha1(vsd)]))


def main():
    os.makedirs(GOLD, exist_ok=True)
    R = _ref_import.load()
    torch.set_num_threads(os.cpu_count())

    # ---------------------------------------------------------------- WaveNet forward
    print("wavenet")
    for tag, cfg, seed, (B, T) in (("small", WN_SMALL, 101, (2, 50)), ("full", WN_FULL, 1234, (2, 96))):
        kw = {k: v for k, v in cfg.items() if k != "dilation_cycle"}
        sd = wavenet_ref.seeded_wavenet_state(seed, **kw)
        net = R["WaveNet"](**cfg).eval()
        net.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, 128, T, generator=g)
        cond = torch.randn(B, 256, T, generator=g)
        t = torch.tensor([37.0, 912.5])[:B]
        masks = torch.zeros(B, T, dtype=torch.bool)
        masks[1, T - T // 4:] = True
        eps = net(x, t, cond)
        eps_masked = net(x, t, cond, x_masks=masks, cond_masks=masks)
        eps_long = net(x, torch.tensor([400], dtype=torch.long), cond)  # naive/plms pass a [1] long tensor
        taps = {}
        mine = wavenet_ref.wavenet_forward(sd, x, t, cond, residual_layers=cfg["residual_layers"],
                                           dilation_cycle=cfg["dilation_cycle"], taps=taps)
        assert torch.equal(mine, eps), "oracle WaveNet != reference"
        assert torch.equal(wavenet_ref.wavenet_forward(sd, x, t, cond, masks, masks, residual_layers=cfg["residual_layers"],
                                                       dilation_cycle=cfg["dilation_cycle"]), eps_masked)
        arrays = dict(x=x, cond=cond, t=t, masks=masks, eps=eps, eps_masked=eps_masked, eps_long=eps_long,
                      x_layer0=taps["x_0"], x_layer_last=taps[f"x_{cfg['residual_layers'] - 1}"],
                      skip_sum=taps["skip_sum"], seed=np.int64(seed), weights_sha1=np.array(state_sha1(sd)))
        if tag == "small":
            arrays.update({"w:" + k: v for k, v in sd.items()})
        save(f"wavenet_{tag}", **arrays)

    # ---------------------------------------------------------------- samplers (small net, every predictor)
    print("samplers (small net)")
    kw = {k: v for k, v in WN_SMALL.items() if k != "dilation_cycle"}
    sd = wavenet_ref.seeded_wavenet_state(101, **kw)
    diff = build_ref_diffusion(R, WN_SMALL, sd)
    den = oracle_denoiser(sd, WN_SMALL)
    B, T = 2, 40
    g = torch.Generator().manual_seed(7)
    feats = torch.randn(B, T, 256, generator=g)
    masks = torch.zeros(B, T, dtype=torch.bool)
    masks[1, 30:] = True
    betas = sampler_ref.beta_schedule()
    for pred, interval, skip in (("unipc", 50, 0), ("unipc", 10, 0), ("plms", 50, 0), ("naive", 50, 0),
                                 ("naive", 1, 900), ("unipc", 100, 400), ("plms", 100, 400)):
        seed = 1000 + interval + skip
        mel0 = None
        if skip:
            mel0 = torch.rand(B, 128, T, generator=g) * -5
        torch.manual_seed(seed)
        ref = diff(feats, sampler_interval=interval, noise_predictor=pred, skip_steps=skip, original_mel=mel0,
                   x_masks=masks, cond_masks=masks)
        # replay the reference's RNG draw order: randn(shape) [or randn_like for q_sample], then per-step randn_like
        torch.manual_seed(seed)
        if skip:
            xs = sampler_ref.norm_spec(mel0, torch.tensor([-5.0]).view(1, 1, -1), torch.tensor([0.0]).view(1, 1, -1))
            x_init = sampler_ref.q_sample(xs, 1000 - skip, torch.randn_like(xs), betas)
        else:
            x_init = torch.randn(B, 128, T)
        n = len(range(0, 1000 - skip, interval))
        step_noise = torch.stack([torch.randn(B, 128, T) for _ in range(n)]) if pred == "naive" else torch.zeros(0)
        mine = sampler_ref.diffusion_sample(den, feats, x_init=x_init, sampler_interval=interval, predictor=pred,
                                            step_noise=step_noise, skip_steps=skip, x_masks=masks, cond_masks=masks)
        assert torch.equal(mine, ref), f"oracle sampler {pred}/{interval}/{skip} != reference"
        save(f"sampler_small_{pred}_i{interval}_s{skip}", features=feats, masks=masks, x_init=x_init,
             step_noise=step_noise, mel=ref, interval=np.int64(interval), skip=np.int64(skip))

    # ---------------------------------------------------------------- BASELINE configs[0] and configs[1] (full net)
    print("samplers (full net): C1 = 5 s / 20-step UniPC, C2 = 10 s / 100-step UniPC")
    kw = {k: v for k, v in WN_FULL.items() if k != "dilation_cycle"}
    sd = wavenet_ref.seeded_wavenet_state(1234, **kw)
    diff = build_ref_diffusion(R, WN_FULL, sd)
    for tag, T, interval, seed in (("c1", 430, 50, 1234), ("c2", 861, 10, 1235)):
        g = torch.Generator().manual_seed(seed)
        feats = torch.randn(1, T, 256, generator=g)
        torch.manual_seed(seed + 100)
        ref = diff(feats, sampler_interval=interval)
        torch.manual_seed(seed + 100)
        x_init = torch.randn(1, 128, T)
        if tag == "c1":
            mine = sampler_ref.diffusion_sample(oracle_denoiser(sd, WN_FULL), feats, x_init=x_init, sampler_interval=interval)
            assert torch.equal(mine, ref)
        save(f"sampler_full_{tag}", features=feats, x_init=x_init, mel=ref, interval=np.int64(interval),
             seed=np.int64(1234), weights_sha1=np.array(state_sha1(sd)))

    # ---------------------------------------------------------------- NSF-HiFiGAN
    print("nsf-hifigan")
    for tag, h, seed, T, store_w in (("v1_small", nsf_hifigan_ref.CONFIG_V1, 55, 24, False),
                                     ("v1_256_small", nsf_hifigan_ref.CONFIG_V1_256, 56, 20, False),
                                     ("v1_full", nsf_hifigan_ref.CONFIG_V1, 55, 861, False)):
        gsd = nsf_hifigan_ref.seeded_generator_state(seed, h)
        gen = R["Generator"](R["AttrDict"](h))
        gen.remove_weight_norm()
        gen.eval()
        gen.load_state_dict(gsd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        mel = torch.randn(1, 128, T, generator=g) * 0.5 - 2.0
        f0 = synth_f0(T, h["sampling_rate"] / h["hop_size"])[None]
        L = T * h["hop_size"]
        torch.manual_seed(seed + 2)
        ref = gen(mel, f0)
        torch.manual_seed(seed + 2)
        rand_ini = torch.rand(1, 9)
        rand_ini[:, 0] = 0
        src_noise = torch.randn(1, L, 9)
        taps = {}
        mine = nsf_hifigan_ref.generator_forward(gsd, h, mel, f0, rand_ini, src_noise, taps)
        assert torch.equal(mine, ref), f"oracle generator {tag} != reference"
        arrays = dict(mel=mel, f0=f0, rand_ini=rand_ini, wav=ref, har_source=taps["har_source"] if T < 100 else torch.zeros(0),
                      stage0=taps["stage_0"] if T < 100 else torch.zeros(0),
                      seed=np.int64(seed), noise_seed=np.int64(seed + 2), weights_sha1=np.array(state_sha1(gsd)),
                      src_noise_sha1=np.array(sha1_of([src_noise])), config=np.array(json.dumps(h)))
        if T < 100:
            arrays["src_noise"] = src_noise
        save(f"nsf_{tag}", **arrays)

    # ---------------------------------------------------------------- STFT / mel
    print("mel")
    pam = R["PitchAdjustableMelSpectrogram"]()
    g = torch.Generator().manual_seed(77)
    n = 44100
    tt = torch.arange(n) / 44100.0
    wav = (0.4 * torch.sin(2 * np.pi * 220 * tt) + 0.2 * torch.sin(2 * np.pi * 1760 * tt + 1.0)
           + 0.05 * torch.randn(n, generator=g))[None]
    arrays = dict(wav=wav)
    for ks, sp in ((0, 1.0), (3, 1.0), (-5, 1.0), (12, 1.0), (0, 1.5)):
        ref = pam(wav, key_shift=ks, speed=sp)
        assert torch.equal(mel_ref.mel_spectrogram(wav, key_shift=ks, speed=sp), ref)
        arrays[f"mel_ks{ks}_sp{sp}"] = ref
    arrays["logmel_log10"] = mel_ref.wav2spec(wav, use_natural_log=False)
    save("mel", **arrays)

    # ---------------------------------------------------------------- condition front end
    print("front end (DiffSinger.forward_features, NaiveProjection encoders)")
    get_mask, fwd_features = R["diffsinger_methods"]()
    Enc = R["NaiveProjectionEncoder"]

    class RefFrontEnd(torch.nn.Module):   # carries the encoders; the two methods are the reference's own source
        forward_features = fwd_features

        def __init__(self, sd):
            super().__init__()
            self.get_mask_from_lengths = get_mask.__func__ if hasattr(get_mask, "__func__") else get_mask
            self.text_encoder = Enc(256, 256)
            self.speaker_encoder = Enc(10, 256, use_embedding=True)
            self.pitch_encoder = Enc(1, 256, preprocessing=R["pitch_to_scale"])
            if "pitch_shift_encoder.projection.weight" in sd:
                self.pitch_shift_encoder = Enc(1, 256)
            if "energy_encoder.projection.weight" in sd:
                self.energy_encoder = Enc(1, 256)
            self.load_state_dict(sd, strict=True)

    g = torch.Generator().manual_seed(99)
    B, T = 3, 70
    contents = torch.randn(B, T, 256, generator=g)
    f0 = torch.rand(B, T, generator=g) * 1300.0            # includes values below f0_min and above f0_max
    f0[1, 20:30] = 0.0
    lens = torch.tensor([70, 51, 64])
    arrays = dict(contents=contents, f0=f0, lens=lens)
    sd_a = features_ref.seeded_frontend_state(11)
    sd_b = features_ref.seeded_frontend_state(12, pitch_shift=True, energy=True)
    ids = torch.tensor([3, 0, 9])
    mix = torch.randn(B, 256, generator=g) * 0.1
    mix_t = torch.randn(B, T, 256, generator=g) * 0.1
    shift = torch.randn(B, 1, generator=g)
    energy = torch.rand(B, T, 1, generator=g)
    cases = {"ids": (sd_a, dict(speakers=ids)), "mix": (sd_a, dict(speakers=mix)), "mix_t": (sd_a, dict(speakers=mix_t)),
             "full": (sd_b, dict(speakers=ids, pitch_shift=shift, energy=energy))}
    for tag, (sd, kw) in cases.items():
        ref = RefFrontEnd(sd).forward_features(kw["speakers"], contents, lens, T, mel_lens=lens, mel_max_len=T, pitches=f0.clone(),
                                               pitch_shift=kw.get("pitch_shift"), energy=kw.get("energy"))
        mine = features_ref.forward_features(sd, contents, kw["speakers"], f0, kw.get("pitch_shift"), kw.get("energy"), lens, T)
        assert torch.equal(mine["features"], ref["features"]) and torch.equal(mine["x_masks"], ref["x_masks"]), tag
        arrays[f"features_{tag}"] = ref["features"]
    arrays.update(ids=ids, mix=mix, mix_t=mix_t, shift=shift, energy=energy, masks=ref["x_masks"],
                  sha1_a=np.array(state_sha1(sd_a)), sha1_b=np.array(state_sha1(sd_b)))
    save("frontend", **arrays)

    # ---------------------------------------------------------------- RefineGAN generator
    print("refinegan")
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fish_diffusion_refinegan_generator", os.path.join(_ref_import.REFERENCE_ROOT, "fish_diffusion/modules/vocoders/refinegan/generator.py"))
    rgmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rgmod)
    for tag, cfg, seed, (B, T) in (("small", dict(refinegan_ref.CONFIG), 21, (2, 5)),
                                   ("hifisinger", dict(refinegan_ref.CONFIG, num_mels=256), 22, (1, 37))):
        gen = rgmod.RefineGANGenerator(**cfg)
        gen.remove_weight_norm()
        gen.eval()
        rsd = refinegan_ref.seeded_state(seed, cfg)
        gen.load_state_dict(rsd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        mel = torch.randn(B, cfg["num_mels"], T, generator=g) * 0.5 - 2.0
        f0 = torch.stack([synth_f0(T, cfg["sampling_rate"] / cfg["hop_length"]) * (1 + 0.3 * b) for b in range(B)])[:, None]
        torch.manual_seed(seed + 2)
        ref = gen(mel, f0)
        torch.manual_seed(seed + 2)                      # replay the reference's randn_like draws, in order
        noises = [torch.randn(sh) for sh in refinegan_ref.noise_shapes(cfg, B, T)]
        taps = {}
        mine = refinegan_ref.generator_forward(rsd, cfg, mel, f0, noises, taps)
        assert torch.equal(mine, ref), f"oracle refinegan {tag} != reference"
        save(f"refinegan_{tag}", mel=mel, f0=f0, wav=ref, template=taps["template"], bottleneck=taps["bottleneck"], up_0=taps["up_0"],
             seed=np.int64(seed), noise_seed=np.int64(seed + 2), weights_sha1=np.array(state_sha1(rsd)),
             noise_sha1=np.array(sha1_of(noises)), config=np.array(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()})))

    # ---------------------------------------------------------------- HiFiSinger (svc_hifisinger_v2): encoders + fuser + RefineGAN
    print("hifisinger")
    get_mask_h, fwd_feat_h, fwd_h = R["hifisinger_methods"]()
    hcfg = dict(refinegan_ref.CONFIG, num_mels=256)

    class RefHiFiSinger(torch.nn.Module):   # encoders + generator are real reference classes, the 3 methods its own source
        forward_features = fwd_feat_h
        forward = fwd_h

        def __init__(self):
            super().__init__()
            self.get_mask_from_lengths = get_mask_h.__func__ if hasattr(get_mask_h, "__func__") else get_mask_h
            self.text_encoder = Enc(768, 256)
            self.speaker_encoder = Enc(10, 256, use_embedding=True)
            self.pitch_shift_encoder = Enc(1, 256)
            self.energy_encoder = Enc(1, 256)
            self.feature_fuser = torch.nn.Sequential(torch.nn.Linear(256, 256), torch.nn.SiLU(), torch.nn.Linear(256, 256), torch.nn.SiLU())
            self.encoder_type = "RefineGAN"
            self.encoder = rgmod.RefineGANGenerator(**hcfg)
            self.encoder.remove_weight_norm()

    hsd = features_ref.seeded_hifisinger_state(8)
    hgsd = refinegan_ref.seeded_state(9, hcfg)
    ref_mod = RefHiFiSinger().eval()
    ref_mod.load_state_dict({**hsd, **{"encoder." + k: v for k, v in hgsd.items()}}, strict=True)
    g = torch.Generator().manual_seed(40)
    B, T = 2, 11
    contents = torch.randn(B, T, 768, generator=g)
    ids = torch.tensor([1, 4])
    lens = torch.tensor([11, 8])
    f0 = torch.stack([synth_f0(T, 44100 / 256), synth_f0(T, 44100 / 256) * 1.5])[:, :, None]
    shift = torch.randn(B, 1, generator=g)
    energy = torch.rand(B, T, 1, generator=g)
    torch.manual_seed(41)
    ref = ref_mod(ids, contents, lens, T, pitches=f0, pitch_shift=shift, energy=energy)
    ref_feat = ref_mod.forward_features(ids, contents, lens, T, pitch_shift=shift, energy=energy)["features"]
    torch.manual_seed(41)
    noises = [torch.randn(sh) for sh in refinegan_ref.noise_shapes(hcfg, B, T)]
    feats = features_ref.hifisinger_features(hsd, contents, ids, lens, T, shift, energy)
    mine = refinegan_ref.generator_forward(hgsd, hcfg, feats["features"].transpose(1, 2), f0.transpose(1, 2), noises)
    assert torch.equal(feats["features"], ref_feat) and torch.equal(mine, ref), "oracle hifisinger != reference"
    save("hifisinger", contents=contents, ids=ids, lens=lens, f0=f0, shift=shift, energy=energy, features=ref_feat, wav=ref,
         noise_seed=np.int64(41), sha1_frontend=np.array(state_sha1(hsd)), sha1_generator=np.array(state_sha1(hgsd)),
         config=np.array(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in hcfg.items()})))

    # HiFiSinger v1 (configs/_base_/archs/hifi_svc.py): same front end (256-dim contents), NSF-HiFiGAN generator with num_mels=256
    h1 = dict(nsf_hifigan_ref.CONFIG_V1, num_mels=256)

    class RefHiFiSingerV1(torch.nn.Module):
        forward_features = fwd_feat_h
        forward = fwd_h

        def __init__(self):
            super().__init__()
            self.get_mask_from_lengths = get_mask_h.__func__ if hasattr(get_mask_h, "__func__") else get_mask_h
            self.text_encoder = Enc(256, 256)
            self.speaker_encoder = Enc(10, 256, use_embedding=True)
            self.pitch_shift_encoder = Enc(1, 256)
            self.energy_encoder = Enc(1, 256)
            self.feature_fuser = torch.nn.Sequential(torch.nn.Linear(256, 256), torch.nn.SiLU(), torch.nn.Linear(256, 256), torch.nn.SiLU())
            self.encoder_type = "HiFiGAN"
            self.encoder = R["Generator"](R["AttrDict"](h1))
            self.encoder.remove_weight_norm()

    hsd1 = features_ref.seeded_hifisinger_state(18, content_dim=256)
    gsd1 = nsf_hifigan_ref.seeded_generator_state(19, h1)
    ref1 = RefHiFiSingerV1().eval()
    ref1.load_state_dict({**hsd1, **{"encoder." + k: v for k, v in gsd1.items()}}, strict=True)
    B, T = 2, 9
    lens = torch.tensor([9, 6])
    f0 = torch.stack([synth_f0(T), synth_f0(T) * 1.5])[:, :, None]
    # RNG order inside Generator: rand_ini (models.py:210) then src_noise (:289)
    g = torch.Generator().manual_seed(50)
    contents = torch.randn(B, T, 256, generator=g)
    energy1 = torch.rand(B, T, 1, generator=g)
    torch.manual_seed(51)
    ref = ref1(ids, contents, lens, T, pitches=f0, pitch_shift=shift, energy=energy1)
    torch.manual_seed(51)
    rand_ini = torch.rand(B, 9)
    rand_ini[:, 0] = 0
    src_noise = torch.randn(B, T * 512, 9)
    feats1 = features_ref.hifisinger_features(hsd1, contents, ids, lens, T, shift, energy1)
    mine = nsf_hifigan_ref.generator_forward(gsd1, h1, feats1["features"].transpose(1, 2), f0[:, :, 0], rand_ini, src_noise)
    assert torch.equal(mine, ref), "oracle hifisinger v1 != reference"
    save("hifisinger_v1", contents=contents, ids=ids, lens=lens, f0=f0, shift=shift, energy=energy1, wav=ref, noise_seed=np.int64(51),
         sha1_frontend=np.array(state_sha1(hsd1)), sha1_generator=np.array(state_sha1(gsd1)), config=np.array(json.dumps(h1)))

    golden_convnext(R)
    golden_frontend_expand(R)
    golden_tfdec(R)
    golden_round2(R)
    golden_convnext_cross(R)
    golden_refinegan_sine(R)
    golden_frontend_svs(R)
    golden_round3(R)
    golden_round4(R)
    golden_round5(R)
    golden_round6(R)

    write_manifest()
    print("done")


if __name__ == "__main__":
    SECTIONS = {"convnext": golden_convnext, "frontend_expand": golden_frontend_expand, "tfdec": golden_tfdec, "round2": golden_round2, "convnext_cross": golden_convnext_cross, "refinegan_sine": golden_refinegan_sine,
                "frontend_svs": golden_frontend_svs, "round3": golden_round3, "round4": golden_round4, "round5": golden_round5, "round6": golden_round6}
    if len(sys.argv) == 2 and sys.argv[1] == "manifest":   # re-index the fixtures on disk (no reference needed)
        write_manifest()
    elif len(sys.argv) == 2 and sys.argv[1] in SECTIONS:   # regenerate one section only
        os.makedirs(GOLD, exist_ok=True)
        torch.set_num_threads(os.cpu_count())
        SECTIONS[sys.argv[1]](_ref_import.load())
        write_manifest()
    else:
        main()
