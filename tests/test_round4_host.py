"""CPU tests added in round 4: the sampler coefficient tables come from the module's BUFFERS (what a checkpoint holds), the oracle with the
same buffers reproduces the reference's mels, `load_checkpoint` + its key-coverage assertion, the `ema_model` preference, and the
provenance of the reference-caller fixture."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import nsf_hifigan_ref, sampler_ref, wavenet_ref
from tests.helpers import GOLD, WN_SMALL, load, rel_err, sha1_state, wavenet_sd

torch.set_num_threads(8)


def _small_diffusion(sd=None):
    from fish_diffusion_amd import GaussianDiffusion
    diff = GaussianDiffusion(dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-5], spec_max=[0])
    if sd is not None:
        diff.denoise_fn.load_state_dict(sd, strict=True)
    return diff.eval()


def _golden_buffers(g):
    naive = {k.split(":", 1)[1]: torch.as_tensor(v) for k, v in g.items() if k.startswith("naive:")}   # (0-dim arrays come back as numpy)
    return naive, g["plms_alphas_cumprod"]


# ------------------------------------------------------------------------------------------------ tables from buffers
def test_sampler_tables_are_built_from_the_predictor_buffers(lib_built):
    """noise_predictor.py:29-71,115: the DDPM / PLMS coefficients are register_buffers -- a checkpoint's values win over the config's."""
    from fish_diffusion_amd import schedule
    diff = _small_diffusion()
    for pred, interval, skip in (("naive", 50, 0), ("naive", 1, 0), ("plms", 50, 0), ("plms", 100, 400), ("naive", 100, 400)):
        kind, tab = diff._sampler_table(pred, interval, skip)
        kind2, ref = schedule.sampler_table(pred, interval=interval, skip_steps=skip, **diff._sched)
        assert kind == kind2 and np.array_equal(tab, np.ascontiguousarray(ref, dtype=np.float32)), (pred, interval, skip)   # untouched buffers: bit-identical
    g = load("sampler_buffers")
    naive, acp = _golden_buffers(g)
    state = {"naive_noise_predictor." + k: v for k, v in naive.items()}
    state["plms_noise_predictor.alphas_cumprod"] = acp
    result = diff.load_state_dict(state, strict=False)
    assert not result.unexpected_keys
    _, tab = diff._sampler_table("naive", 50, 0)
    chunks = schedule.timestep_chunks(1000, 0, 50)
    assert [int(r[0]) for r in tab] == chunks
    for r, t in zip(tab, chunks):
        assert r[1] == float(naive["sqrt_recip_alphas_cumprod"][t]) and r[2] == float(naive["sqrt_recipm1_alphas_cumprod"][t])
        assert r[3] == float(naive["posterior_mean_coef1"][t]) and r[4] == float(naive["posterior_mean_coef2"][t])
        want = (0.5 * naive["posterior_log_variance_clipped"][t]).exp() if t > 0 else torch.tensor(0.0)
        assert r[5] == float(want)
        assert r[6] == np.float32(-0.9) and r[7] == np.float32(0.8)
    _, ref = schedule.sampler_table("naive", interval=50, skip_steps=0, **diff._sched)
    assert not np.array_equal(tab, ref)                                       # the loaded buffers changed the rows
    _, ptab = diff._sampler_table("plms", 50, 0)
    a_t, a_p = acp[chunks[3]], acp[max(chunks[3] - 50, 0)]
    assert ptab[3][2] == float(a_p - a_t)
    # in-place edits bump the tensor version: the cache must not serve stale rows
    diff.naive_noise_predictor.clip_max.fill_(0.5)
    assert diff._sampler_table("naive", 50, 0)[1][0][7] == np.float32(0.5)
    # UniPC's schedule is NOT a buffer in the reference (noise_predictor.py:151-158): it follows the constructor
    k, utab = diff._sampler_table("unipc", 50, 0)
    assert np.array_equal(utab, np.ascontiguousarray(schedule.sampler_table("unipc", interval=50, **diff._sched)[1], dtype=np.float32))
    with pytest.raises(ValueError):
        diff.plms_noise_predictor.alphas_cumprod = torch.ones(10)
        diff._sampler_table("plms", 50, 0)


def test_oracle_with_loaded_buffers_reproduces_the_reference_mels():
    g = load("sampler_buffers")
    sd = wavenet_sd(WN_SMALL, int(g["weights_seed"]))
    assert sha1_state(sd) == str(g["weights_sha1"])
    naive, acp = _golden_buffers(g)
    den = lambda x, t, c, xm, cm: wavenet_ref.wavenet_forward(sd, x, t, c, xm, cm, residual_layers=4, dilation_cycle=4)   # noqa: E731
    with torch.no_grad():
        mel = sampler_ref.diffusion_sample(den, g["features"], x_init=g["x_naive"], sampler_interval=int(g["interval"]), predictor="naive",
                                           step_noise=g["step_noise"], naive_buffers=naive)
        assert rel_err(mel, g["mel_naive"]) < 1e-5
        plain = sampler_ref.diffusion_sample(den, g["features"], x_init=g["x_naive"], sampler_interval=int(g["interval"]), predictor="naive",
                                             step_noise=g["step_noise"])
        assert rel_err(plain, g["mel_naive"]) > 1e-2                           # the schedule-derived coefficients give a different mel
        mel = sampler_ref.diffusion_sample(den, g["features"], x_init=g["x_plms"], sampler_interval=int(g["interval"]), predictor="plms",
                                           plms_alphas_cumprod=acp)
        assert rel_err(mel, g["mel_plms"]) < 1e-5


# ------------------------------------------------------------------------------------------------ load_checkpoint
class Cfg(dict):
    """mmengine.Config stand-in: attribute access + .get on nested dicts."""
    __getattr__ = dict.get

    def __init__(self, d):
        super().__init__({k: (Cfg(v) if isinstance(v, dict) else v) for k, v in d.items()})

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Cfg) else v) for k, v in self.items()}


def svc_config(tmp_path, ema=0.999, wn=WN_SMALL):
    from fish_diffusion_amd import pitch_to_scale
    from fish_diffusion_amd.nsf_hifigan import Generator
    h = dict(nsf_hifigan_ref.CONFIG_V1)
    (tmp_path / "config.json").write_text(json.dumps(h))
    if not (tmp_path / "voc_model").exists():
        torch.save({"generator": Generator(h).state_dict()}, tmp_path / "voc_model")       # weight-norm form, like the released checkpoints
    model = dict(type="DiffSinger",
                 text_encoder=dict(type="NaiveProjectionEncoder", input_size=256, output_size=256),
                 speaker_encoder=dict(type="NaiveProjectionEncoder", input_size=10, output_size=256, use_embedding=True),
                 pitch_encoder=dict(type="NaiveProjectionEncoder", input_size=1, output_size=256, preprocessing=pitch_to_scale),
                 diffusion=dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", **wn), spec_min=[-5], spec_max=[0]),
                 vocoder=dict(type="NsfHifiGAN", checkpoint_path=str(tmp_path / "voc_model"), use_natural_log=False))
    return Cfg(dict(model=model, ema_momentum=ema))


def lightning_checkpoint(model, fe_model, fe_ema, wn_model, wn_ema):
    """A Lightning-style checkpoint for SVCModel: model.* / ema_model.* (front end + denoiser weights from the given states, the diffusion
    buffers from the module), plus vocoder.* entries the loader must drop."""
    sd = {}
    for prefix, fe, wn in (("model.", fe_model, wn_model), ("ema_model.", fe_ema, wn_ema)):
        if not hasattr(model, prefix[:-1]):
            continue
        for k, v in getattr(model, prefix[:-1]).state_dict().items():
            if k.startswith("diffusion.denoise_fn."):
                sd[prefix + k] = wn[k[len("diffusion.denoise_fn."):]].clone()
            elif k in fe:
                sd[prefix + k] = fe[k].clone()
            else:
                sd[prefix + k] = v.clone()           # buffers: betas, alphas_cumprod, spec_min, predictor tables ...
    sd["vocoder.model.conv_pre.weight"] = torch.zeros(3)   # wrong shape on purpose: must be dropped, not loaded
    return {"state_dict": sd, "epoch": 3}


def test_load_checkpoint_covers_every_key_and_prefers_the_ema_model(lib_built, tmp_path):
    from fish_diffusion_amd.inference import SVCModel, inference_model, load_checkpoint
    from oracle import features_ref
    cfg = svc_config(tmp_path)
    fe_m, fe_e = features_ref.seeded_frontend_state(81), features_ref.seeded_frontend_state(82)
    wn_m, wn_e = wavenet_sd(WN_SMALL, 83), wavenet_sd(WN_SMALL, 84)
    ck = lightning_checkpoint(SVCModel(cfg), fe_m, fe_e, wn_m, wn_e)
    torch.save(ck, tmp_path / "step_000100.ckpt")
    report = {}
    m = load_checkpoint(cfg, str(tmp_path / "step_000100.ckpt"), device="cpu", report=report)
    assert report["missing"] == [] and report["unexpected"] == [] and not m.training
    assert inference_model(m) is m.ema_model                                  # tools/diffusion/inference.py:134-138
    assert torch.equal(m.ema_model.diffusion.denoise_fn.state_dict()["input_projection.conv.weight"], wn_e["input_projection.conv.weight"])
    assert torch.equal(m.model.text_encoder.projection.weight, fe_m["text_encoder.projection.weight"])
    assert not any(p.requires_grad for p in m.vocoder.parameters())
    # a directory: the naturally-sorted last checkpoint (tools/diffusion/inference.py:67-74)
    ckdir = tmp_path / "ckpts"
    ckdir.mkdir()
    torch.save(ck, ckdir / "step_9.ckpt")
    bad = {"state_dict": {k: v for k, v in ck["state_dict"].items() if "residual_layers.2.output_projection" not in k}}
    torch.save(bad, ckdir / "step_10.ckpt")                                  # natural order: 10 after 9
    with pytest.raises(KeyError, match="residual_layers.2.output_projection"):
        load_checkpoint(cfg, str(ckdir), device="cpu")
    rep = {}
    load_checkpoint(cfg, str(ckdir), device="cpu", allow_missing=True, report=rep)      # the reference's strict=False behaviour, on request
    assert len(rep["missing"]) == 4 and all("residual_layers.2.output_projection" in k for k in rep["missing"])
    # no ema_model.* in a checkpoint whose config builds one: refused
    no_ema = {"state_dict": {k: v for k, v in ck["state_dict"].items() if not k.startswith("ema_model.")}}
    with pytest.raises(KeyError, match="ema_model"):
        load_checkpoint(cfg, no_ema, device="cpu")
    # without ema_momentum the caller falls back to .model
    cfg2 = svc_config(tmp_path, ema=None)
    m2 = load_checkpoint(cfg2, no_ema, device="cpu", report=rep)
    assert not hasattr(m2, "ema_model") and inference_model(m2) is m2.model and rep["unexpected"] == []
    # unexpected keys (discriminators, optimiser shards) are reported, not fatal -- like strict=False
    extra = dict(no_ema["state_dict"], **{"mpd.discriminators.0.weight": torch.zeros(1)})
    load_checkpoint(cfg2, {"state_dict": extra}, device="cpu", report=rep)
    assert rep["unexpected"] == ["mpd.discriminators.0.weight"]
    with pytest.raises(NotImplementedError):
        SVCModel(Cfg(dict(cfg2.to_dict(), lora=True)))


# ------------------------------------------------------------------------------------------------ the reference-caller fixture
def test_reference_caller_fixture_provenance():
    """tests/golden/svc_inference_forward.json holds the source of ONE reference method (test infrastructure for the GPU box, which has no
    reference tree).  Where the reference tree exists, the text must be exactly those lines of that file."""
    with open(os.path.join(GOLD, "svc_inference_forward.json")) as f:
        fx = json.load(f)
    assert fx["reference_file"] == "tools/diffusion/inference.py" and "def forward(" in fx["source"]
    compile(fx["source"], "svc_inference_forward", "exec")
    path = os.path.join(os.environ.get("FISH_REFERENCE_ROOT", "/root/reference"), fx["reference_file"])
    if not os.path.exists(path):
        pytest.skip("no reference tree on this box")
    with open(path) as f:
        text = f.read()
    assert hashlib.sha256(text.encode()).hexdigest() == fx["file_sha256"]
    a, b = fx["lines"]
    lines = text.splitlines()[a - 1:b]
    indent = len(lines[0]) - len(lines[0].lstrip())
    assert "\n".join(ln[indent:] for ln in lines) + "\n" == fx["source"]
    g = load("svc_caller")
    assert g["wav"].shape == (70 * 512,) and int(g["n_audio"]) // 512 == 70


def test_branch_free_gelu_rational_is_fp32_class():
    """`erf_f` / `gelu_f` in csrc/convgemm.hip.h (the GELU epilogue of ConvNext's pwconv1 and the transformer's linear1): the coefficients are
    read from the header itself, evaluated in float32 the way the kernel does (Horner, fma order aside), and held to the accuracy the comment
    there states -- erf within 5e-7 abs, GELU within 1.5e-6 abs of an fp64 evaluation on [-6, 6] (torch's own fp32 gelu: 1.2e-6)."""
    import re
    from scipy.special import erf
    src = open(os.path.join(os.path.dirname(__file__), "..", "fish_diffusion_amd", "csrc", "convgemm.hip.h")).read()
    body = src[src.index("float erf_f(float x) {"):src.index("float gelu_f(float x)")]
    p0, q0 = (float(v) for v in re.search(r"float p = (\S+?)f, q = (\S+?)f;", body).groups())
    ps = [p0] + [float(v) for v in re.findall(r"p = fmaf\(p, x2, (\S+?)f\);", body)]
    qs = [q0] + [float(v) for v in re.findall(r"q = fmaf\(q, x2, (\S+?)f\);", body)]
    assert len(ps) == 7 and len(qs) == 5

    def erf32(x):
        x = np.clip(x, -4, 4).astype(np.float32)
        x2 = (x * x).astype(np.float32)
        p = np.float32(ps[0]); q = np.float32(qs[0])
        for c in ps[1:]:
            p = (p * x2 + np.float32(c)).astype(np.float32)
        for c in qs[1:]:
            q = (q * x2 + np.float32(c)).astype(np.float32)
        r = ((x * p / q).astype(np.float32) * np.float32(rcp_err)).astype(np.float32)                   # v_rcp_f32 is good to 1 ulp
        return np.where(x2 >= 16, np.copysign(np.float32(1), x), r).astype(np.float32)

    assert "return x2 >= 16.f ? copysignf(1.f, x) : r;" in body
    # the tails: with the result clamped, gelu is exactly 0 / x beyond the input clamp whatever the reciprocal's last bit does
    for rcp_err in (1 - 2.0 ** -22, 1.0, 1 + 2.0 ** -22):
        xt = np.concatenate([np.linspace(-1e4, -5.66, 20001), np.linspace(5.66, 1e4, 20001)]).astype(np.float32)
        gt = (np.float32(0.5) * xt * (1 + erf32((xt * np.float32(0.70710678118654752440)).astype(np.float32)))).astype(np.float32)
        assert np.all(gt[xt < 0] == 0) and np.all(gt[xt > 0] == xt[xt > 0]), rcp_err
    rcp_err = 1.0
    x = np.linspace(-6, 6, 400001).astype(np.float32)
    assert np.abs(erf32(x) - erf(x.astype(np.float64))).max() < 5e-7
    g64 = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    g32 = (np.float32(0.5) * x * (1 + erf32((x * np.float32(0.70710678118654752440)).astype(np.float32)))).astype(np.float32)
    assert np.abs(g32 - g64).max() < 1.5e-6
    assert np.abs(torch.nn.functional.gelu(torch.from_numpy(x)).numpy() - g64).max() > 1.0e-6   # the bar is the reference's own fp32 class


def test_ragged_noise_scatter_matches_the_per_item_copies():
    """`GaussianDiffusion._ragged_scatter` (one gather + one index_copy per chunk) against the per-step, per-item copies it replaced."""
    from fish_diffusion_amd.diffusion import GaussianDiffusion
    g = torch.Generator().manual_seed(3)
    lens, T, M, c, gap = [70, 33, 64, 1], 70, 5, 4, 16
    offs, cur = [], 0
    for n in lens:
        offs.append(cur)
        cur += n + gap
    Tc = cur
    src = torch.randn(c, len(lens), M, T, generator=g)
    want = torch.zeros(c, M, Tc)
    for i in range(c):
        for b, (o, n) in enumerate(zip(offs, lens)):
            want[i, :, o:o + n] = src[i, b, :, :n]
    cols, flat = GaussianDiffusion._ragged_index(offs, lens, T, torch.device("cpu"))
    got = torch.zeros(c, 1, M, Tc)
    GaussianDiffusion._ragged_scatter(got[:, 0], src, cols, flat)
    assert torch.equal(got[:, 0], want)
    one = torch.zeros(c, 1, M, Tc)
    for i in range(c):
        GaussianDiffusion._ragged_scatter(one[i:i + 1, 0], src[i][None], cols, flat)
    assert torch.equal(one, got)
