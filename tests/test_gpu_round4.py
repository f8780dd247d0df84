"""GPU tests added in round 4: dilations beyond the LDS-window kernels' pad, the sampler tables built from the module's BUFFERS,
the checkpoint loader, the reference's own SVCInference.forward body over the installed modules, the fused small-channel ResBlock
kernel against the per-conv path, and the fused per-step seam of the UniPC loop."""
import os

import pytest
import torch

from tests.helpers import ROOT, WN_SMALL, abs_err, load, rel_err, sha1_state, wavenet_sd  # noqa: F401

pytestmark = pytest.mark.gpu

MEL_REL = 1e-3   # north_star: 1e-3 rel fp32 on mel
WAV_ABS = 1e-4   # north_star: 1e-4 abs on waveform samples


@pytest.fixture(scope="module")
def dev(lib_built):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


# ------------------------------------------------------------------------------------------------ dilation 16
def test_dilation_cycle_5_fp32_parity_and_lds_window_modes(dev):
    """`dilation_cycle=5` gives a layer with dilation 16.  The fp32 kernels read their taps from the padded rows (32-column halo) and must
    match the oracle; the fp16-split kernels stage tile +/- 8 columns in LDS, so `storage = "fp16x3"` must REFUSE such a net
    (NotImplementedError) instead of reading outside its window; bf16 storage must route around its LDS-tiled kernels (register-direct
    kernels for every geometry) and stay inside its usual error regime."""
    from fish_diffusion_amd import DENOISERS
    from oracle import wavenet_ref
    cfg = dict(WN_SMALL, residual_channels=64, residual_layers=5, dilation_cycle=5)
    sd = wavenet_sd(cfg, 41)
    net = DENOISERS.build(dict(type="WaveNetDenoiser", **cfg))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    g = torch.Generator().manual_seed(3)
    for B, T in ((1, 97), (2, 300)):
        x, c, t = torch.randn(B, 128, T, generator=g), torch.randn(B, 256, T, generator=g), (torch.rand(B, generator=g) * 999)
        with torch.no_grad():
            ref = wavenet_ref.wavenet_forward(sd, x, t, c, None, None, residual_layers=5, dilation_cycle=5)
        net.storage = "fp32"
        y = net(x.to(dev), t.to(dev), c.to(dev)).cpu()
        assert rel_err(y, ref) < 2e-5, (B, T, rel_err(y, ref))
        net.storage = "bf16"
        yb = net(x.to(dev), t.to(dev), c.to(dev)).cpu()
        assert torch.isfinite(yb).all() and rel_err(yb, ref) < 3e-2, (B, T, rel_err(yb, ref))
        net.storage = "fp32"
    with pytest.raises(NotImplementedError):
        net.storage = "fp16x3"
        net(x.to(dev), t.to(dev), c.to(dev))
    net.storage = "fp32"
    y2 = net(x.to(dev), t.to(dev), c.to(dev)).cpu()
    assert torch.equal(y, y2)                      # the refusal left the handle in a usable fp32 state


# ------------------------------------------------------------------------------------------------ sampler tables from the buffers
def test_samplers_use_the_loaded_predictor_buffers(dev):
    """A checkpoint whose predictor buffers differ from what its config would compute (noise_predictor.py:29-71,115 are register_buffers):
    the reference samples with the checkpoint's values.  `sampler_buffers` holds such buffers and the REAL reference's naive / PLMS mels."""
    from fish_diffusion_amd import GaussianDiffusion
    g = load("sampler_buffers")
    sd = wavenet_sd(WN_SMALL, int(g["weights_seed"]))
    assert sha1_state(sd) == str(g["weights_sha1"])
    diff = GaussianDiffusion(dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-5], spec_max=[0])
    diff.denoise_fn.load_state_dict(sd, strict=True)
    state = {"naive_noise_predictor." + k.split(":", 1)[1]: torch.as_tensor(v) for k, v in g.items() if k.startswith("naive:")}
    state["plms_noise_predictor.alphas_cumprod"] = g["plms_alphas_cumprod"]
    assert not diff.load_state_dict(state, strict=False).unexpected_keys
    diff = diff.to(dev).eval()
    iv = int(g["interval"])
    mel = diff(g["features"].to(dev), sampler_interval=iv, noise_predictor="naive", x_init=g["x_naive"].to(dev), step_noise=g["step_noise"].to(dev))
    e1 = rel_err(mel.cpu(), g["mel_naive"])
    mel = diff(g["features"].to(dev), sampler_interval=iv, noise_predictor="plms", x_init=g["x_plms"].to(dev))
    e2 = rel_err(mel.cpu(), g["mel_plms"])
    print(f"samplers on loaded buffers vs the reference: naive {e1:.3e}, plms {e2:.3e}")
    assert e1 < MEL_REL and e2 < MEL_REL


# ------------------------------------------------------------------------------------------------ the reference's caller over our modules (6b, 6c)
def test_reference_caller_body_runs_over_the_installed_modules(dev, tmp_path):
    """`SVCInference.forward` (tools/diffusion/inference.py:86-162) -- the reference's OWN source text for that one method, extracted by
    oracle/make_golden.py round4 into tests/golden/svc_inference_forward.json -- executed over the MI355X modules: `load_checkpoint` (ours:
    Lightning state dict, vocoder.* dropped, coverage asserted) -> `.ema_model` preference -> `forward_features` -> `diffusion(...)` ->
    `vocoder.spec2wav(result[0].T, f0=pitches)`, with stub extractors.  The golden is the same body over the reference's own CPU modules.
    The global-RNG draws (x_T, the vocoder's rand_ini and source noise) are fed in the reference's order through torch.randn / torch.rand."""
    import json
    from typing import Optional
    from unittest import mock
    import numpy as np
    from fish_diffusion_amd import repeat_expand
    from fish_diffusion_amd.inference import SVCModel, load_checkpoint
    from oracle import features_ref, nsf_hifigan_ref
    from tests.test_round4_host import lightning_checkpoint, svc_config
    g = load("svc_caller")
    with open(os.path.join(ROOT, "tests", "golden", "svc_inference_forward.json")) as f:
        fx = json.load(f)
    ns = {"torch": torch, "np": np, "Optional": Optional, "repeat_expand": repeat_expand}
    exec(compile(fx["source"], "reference:tools/diffusion/inference.py", "exec"), ns)

    s_fe_m, s_fe_e, s_wn_m, s_wn_e, s_voc = [int(v) for v in g["seeds"]]
    fe_m, fe_e = features_ref.seeded_frontend_state(s_fe_m), features_ref.seeded_frontend_state(s_fe_e)
    wn_m, wn_e = wavenet_sd(WN_SMALL, s_wn_m), wavenet_sd(WN_SMALL, s_wn_e)
    hv = nsf_hifigan_ref.CONFIG_V1
    vsd = nsf_hifigan_ref.seeded_generator_state(s_voc, hv)
    assert [sha1_state(x) for x in (fe_m, fe_e, wn_m, wn_e, vsd)] == [str(v) for v in g["sha1"]]
    cfg = svc_config(tmp_path)
    torch.save(lightning_checkpoint(SVCModel(cfg), fe_m, fe_e, wn_m, wn_e), tmp_path / "model.ckpt")
    report = {}
    lm = load_checkpoint(cfg, str(tmp_path / "model.ckpt"), device=dev, report=report)
    assert report["missing"] == [] and report["unexpected"] == []
    lm.vocoder.model.load_folded_state(vsd)            # (the released vocoder ships as its own file; here the seeded generator of the fixture)
    lm.vocoder.to(dev)

    text_features, pitches = g["text_features"].to(dev), g["pitches"].to(dev)

    class Holder(torch.nn.Module):                       # the attributes SVCInference.forward touches (inference.py:49-84)
        forward = ns["forward"]

        def __init__(self):
            super().__init__()
            self.config = cfg
            self.model = lm
            self.text_features_extractor = lambda audio, sr: text_features.clone()
            self.pitch_extractor = lambda audio, sr, pad_to=None: pitches[:pad_to].clone()

        @property
        def device(self):
            return next(self.parameters()).device

    holder = Holder().eval()
    T = int(g["n_audio"]) // 512
    torch.manual_seed(int(g["noise_seed"]))
    draws_n = [torch.randn(1, 128, T)]
    draws_u = [torch.rand(1, 9)]
    draws_n.append(torch.randn(1, T * 512, 9))
    real_randn, real_rand = torch.randn, torch.rand

    def fake_randn(*size, **kw):
        want = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        t = draws_n.pop(0)
        assert tuple(t.shape) == want, (tuple(t.shape), want)
        return t.to(kw.get("device", "cpu"))

    def fake_rand(*size, **kw):
        t = draws_u.pop(0)
        return t.to(kw.get("device", "cpu"))

    audio = torch.zeros(1, int(g["n_audio"]), device=dev)
    with mock.patch("torch.randn", fake_randn), mock.patch("torch.rand", fake_rand):
        wav = holder(audio, 44100, pitch_adjust=int(g["pitch_adjust"]), speakers=torch.tensor([int(g["speaker"])]),
                     sampler_interval=int(g["sampler_interval"]))
    assert torch.randn is real_randn and torch.rand is real_rand and not draws_n and not draws_u
    wav = torch.from_numpy(np.asarray(wav))
    err = abs_err(wav, g["wav"])
    print(f"SVCInference.forward over the installed modules vs over the reference's: wav abs err {err:.3e}")
    assert wav.shape == g["wav"].shape and err < WAV_ABS


# ------------------------------------------------------------------------------------------------ fused ResBlock1 (small-channel stages)
_VOC_CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from fish_diffusion_amd import NsfHifiGAN
from oracle import nsf_hifigan_ref
from tests.helpers import synth_f0
dev = torch.device("cuda", 0)
out = {}
for tag, h in (("v1", nsf_hifigan_ref.CONFIG_V1), ("v1_256", nsf_hifigan_ref.CONFIG_V1_256)):
    gsd = nsf_hifigan_ref.seeded_generator_state(55, h)
    voc = NsfHifiGAN.from_state(h, gsd).to(dev)
    g = torch.Generator().manual_seed(9)
    for B, T in ((1, 3), (2, 37), (1, 300)):
        m = torch.randn(B, 128, T, generator=g) * 0.5 - 2.0
        f0 = torch.stack([synth_f0(T, h["sampling_rate"] / h["hop_size"]) * (1 + 0.2 * b) for b in range(B)])
        ri = torch.rand(B, 9, generator=g)
        ri[:, 0] = 0
        sn = torch.randn(B, T * h["hop_size"], 9, generator=g)
        wav = voc.model(m.to(dev), f0.to(dev), rand_ini=ri.to(dev), src_noise=sn.to(dev))
        out[f"{tag}_{B}_{T}"] = wav.cpu().numpy()
np.savez(sys.argv[1], **out)
print("OK")
'''


def test_fused_resblock_kernel_agrees_with_the_per_conv_path_and_the_oracle(dev, tmp_path):
    """csrc/resblock_fused.hip.h: the C = 16 / 32 stages' ResBlock1 as ONE launch (six convs out of LDS, halo recomputed per tile) against
    (a) the same generator run conv by conv (FDX_NSF_FUSED=0; own process: the switch is read once) and (b) the CPU oracle, on
    geometries that straddle tiles: a signal shorter than one tile, batch 2, 300 frames = 153 600 samples (many tiles, ragged last tile),
    both shipped configs (hop 512: stages of 32 and 16 channels; hop 256: 32)."""
    import subprocess
    import sys
    import numpy as np
    from oracle import nsf_hifigan_ref
    from tests.helpers import synth_f0
    paths = {}
    for mode in ("1", "0"):
        paths[mode] = str(tmp_path / f"voc_{mode}.npz")
        env = dict(os.environ, FDX_NSF_FUSED=mode)
        r = subprocess.run([sys.executable, "-c", _VOC_CODE % ROOT, paths[mode]], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    fused, plain = np.load(paths["1"]), np.load(paths["0"])
    worst = 0.0
    for k in fused.files:
        d = float(np.abs(fused[k].astype(np.float64) - plain[k]).max())
        worst = max(worst, d)
        assert fused[k].shape == plain[k].shape and np.isfinite(fused[k]).all() and d < 2e-5, (k, d)
    # the oracle on the mid-size case of each config (the 10 s goldens of tests/test_gpu_parity.py / round2 run the fused path as well)
    for tag, h in (("v1", nsf_hifigan_ref.CONFIG_V1), ("v1_256", nsf_hifigan_ref.CONFIG_V1_256)):
        gsd = nsf_hifigan_ref.seeded_generator_state(55, h)
        g = torch.Generator().manual_seed(9)
        for B, T in ((1, 3), (2, 37)):
            m = torch.randn(B, 128, T, generator=g) * 0.5 - 2.0
            f0 = torch.stack([synth_f0(T, h["sampling_rate"] / h["hop_size"]) * (1 + 0.2 * b) for b in range(B)])
            ri = torch.rand(B, 9, generator=g)
            ri[:, 0] = 0
            sn = torch.randn(B, T * h["hop_size"], 9, generator=g)
            with torch.no_grad():
                ref = nsf_hifigan_ref.generator_forward(gsd, h, m, f0, ri, sn)
            err = abs_err(torch.from_numpy(fused[f"{tag}_{B}_{T}"]), ref)
            assert err < WAV_ABS, (tag, B, T, err)
    print(f"fused ResBlock1 vs per-conv path: max abs diff {worst:.3e}")


# ------------------------------------------------------------------------------------------------ UniPC corrector in the last projection's epilogue
_UNIPC_CODE = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from fish_diffusion_amd import GaussianDiffusion
from tests.helpers import WN_SMALL, wavenet_sd
dev = torch.device("cuda", 0)
diff = GaussianDiffusion(dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-5], spec_max=[0])
diff.denoise_fn.load_state_dict(wavenet_sd(WN_SMALL, 101), strict=True)
diff = diff.to(dev).eval()
h = hashlib.sha1()
g = torch.Generator().manual_seed(21)
for B, T, iv in ((1, 37, 100), (2, 113, 50), (1, 430, 20), (3, 64, 500), (1, 861, 334)):
    feats, x0 = torch.randn(B, T, 256, generator=g).to(dev), torch.randn(B, 128, T, generator=g).to(dev)
    m = torch.zeros(B, T, dtype=torch.bool, device=dev)
    m[-1, T - T // 4:] = True
    for masks in (None, m):
        mel = diff(feats, sampler_interval=iv, x_init=x0, x_masks=masks, cond_masks=masks)
        assert torch.isfinite(mel).all()
        h.update(mel.cpu().numpy().tobytes())
    lens = [max(1, T - 7 * b) for b in range(B)]
    mel = diff(feats, sampler_interval=iv, x_init=x0, lengths=lens)        # exact-ragged rows run the same fused epilogue
    h.update(mel.cpu().numpy().tobytes())
# round 6: the other two denoisers' last projection takes the same epilogue
from tests.helpers import CN_SMALL, TD_SMALL, convnext_sd, tfdec_sd
for kind, cfg, sd in (("ConvNextDenoiser", CN_SMALL, convnext_sd(CN_SMALL, 5)), ("TransformerDecoderDenoiser", TD_SMALL, tfdec_sd(TD_SMALL, 6))):
    d2 = GaussianDiffusion(dict(type=kind, **cfg), spec_min=[-5], spec_max=[0])
    d2.denoise_fn.load_state_dict(sd, strict=True)
    d2 = d2.to(dev).eval()
    for B, T, iv in ((1, 37, 100), (2, 113, 334)):
        feats, x0 = torch.randn(B, T, 256, generator=g).to(dev), torch.randn(B, 128, T, generator=g).to(dev)
        m = torch.zeros(B, T, dtype=torch.bool, device=dev)
        m[-1, T - T // 4:] = True
        for masks in (None, m):
            mel = d2(feats, sampler_interval=iv, x_init=x0, x_masks=masks, cond_masks=masks)
            assert torch.isfinite(mel).all()
            h.update(mel.cpu().numpy().tobytes())
        mel = d2(feats, sampler_interval=iv, x_init=x0, lengths=[max(1, T - 7 * b) for b in range(B)])
        h.update(mel.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
'''


def test_unipc_update_fused_into_the_last_projection_is_bit_identical(dev):
    """VERDICT r3 item 3 (first half): the UniPC corrector + next predictor (uni_pc.py:664-680) run in the epilogue of the denoiser's last
    projection (EpiUniPC) instead of a separate elementwise launch.  Same expressions, same order, -ffp-contract=off: the sampled mels must
    be BIT-identical to the unfused sequence (FDX_UNIPC_FUSED=0: out-projection -> eps, then k_unipc_post_pre), for odd step counts (the
    order-1 tail), masks, batches and exact-ragged rows."""
    import subprocess
    import sys
    digests = {}
    for mode in ("1", "0"):
        env = dict(os.environ, FDX_UNIPC_FUSED=mode)
        r = subprocess.run([sys.executable, "-c", _UNIPC_CODE % ROOT], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "DIGEST" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
        digests[mode] = r.stdout.split("DIGEST")[1].split()[0]
    assert digests["1"] == digests["0"], digests
