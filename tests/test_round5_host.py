"""CPU (round 5): the loose ends of VERDICT r4 that need no GPU -- ResBlock2 pinned against the real reference, the slaney filterbank
held to a third-party table, per-utterance failure isolation, checkpoint-coverage rules."""
import json

import numpy as np
import pytest
import torch

from oracle import mel_ref, nsf_hifigan_ref
from tests.helpers import abs_err, load, sha1_state


@pytest.mark.parametrize("tag", ["small", "long"])
def test_resblock2_oracle_matches_reference(tag):
    """models.py:119-158: the fixture is the REAL `Generator(resblock="2")`'s waveform.  Its `leaky_relu(x, inplace=True)` (:152) makes the
    residual the activated x and leaks the activation into the next ResBlock2 of the stage; the restatement has to reproduce both."""
    g = load(f"nsf_rb2_{tag}")
    h = json.loads(str(g["config"]))
    assert h["resblock"] == "2"
    sd = nsf_hifigan_ref.seeded_generator_state(int(g["seed"]), h)
    assert sha1_state(sd) == str(g["weights_sha1"])
    B, T = g["mel"].shape[0], g["mel"].shape[-1]
    torch.manual_seed(int(g["noise_seed"]))
    rand_ini = torch.rand(B, 9)
    rand_ini[:, 0] = 0
    src_noise = torch.randn(B, T * h["hop_size"], 9)
    assert torch.equal(rand_ini, g["rand_ini"])
    mel_in = g["mel"].clone()
    with torch.no_grad():
        wav = nsf_hifigan_ref.generator_forward(sd, h, g["mel"], g["f0"], rand_ini, src_noise)
    assert torch.equal(g["mel"], mel_in), "the in-place activation must not reach the caller's mel"
    assert abs_err(wav, g["wav"]) < 1e-5


def test_slaney_filterbank_against_third_party_table(lib_built):
    """pitch_adjustable_mel.py:44-53 -> librosa.filters.mel.  librosa is not installable here; the table in the fixture comes from
    HuggingFace transformers' `audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` (its own tests hold it to librosa) --
    an implementation neither the product (`fdx_mel_filterbank`) nor `oracle/mel_ref.py` shares a line with."""
    from fish_diffusion_amd import PitchAdjustableMelSpectrogram
    g = load("mel_filterbank_hf")
    assert "transformers" in str(g["source"])
    for n_fft in (int(n) for n in g["n_ffts"]):
        hf = g[f"fb_nfft{n_fft}"].numpy()
        peak = np.abs(hf).max()
        ours = mel_ref.slaney_mel_filterbank(sr=44100, n_fft=n_fft, n_mels=128, fmin=40, fmax=16000)
        assert np.abs(ours - hf).max() < 5e-7 * peak, n_fft
        prod = PitchAdjustableMelSpectrogram(n_fft=n_fft, win_length=n_fft).filterbank().numpy()
        assert prod.shape == hf.shape and np.abs(prod - hf).max() < 5e-7 * peak, n_fft


# ------------------------------------------------------------------------------------------------ load_checkpoint
class _Opaque:                       # a class the weights-only unpickler does not allow-list (stands in for optimizer / callback state)
    def __init__(self):
        self.lr = 1e-3


def test_load_checkpoint_schedule_buffers_warn_and_full_unpickler_fallback(lib_built, tmp_path):
    """A reference checkpoint saved without the predictor sub-modules' buffers loads in the reference (strict=False) and samples with the
    schedule `__init__` computes from the config: here that is a warning + `report["missing_schedule_buffers"]`, while a missing weight or
    `spec_min` still raises.  A Lightning file whose extra state needs the full unpickler loads with a warning unless `weights_only=True`."""
    from fish_diffusion_amd.inference import SVCModel, load_checkpoint
    from oracle import features_ref
    from tests.helpers import WN_SMALL, wavenet_sd
    from tests.test_round4_host import lightning_checkpoint, svc_config
    cfg = svc_config(tmp_path)
    ck = lightning_checkpoint(SVCModel(cfg), features_ref.seeded_frontend_state(81), features_ref.seeded_frontend_state(82),
                              wavenet_sd(WN_SMALL, 83), wavenet_sd(WN_SMALL, 84))
    sd = ck["state_dict"]
    no_sched = {k: v for k, v in sd.items() if "noise_predictor." not in k and not k.endswith(("diffusion.betas", "alphas_cumprod"))}
    assert len(no_sched) < len(sd)
    rep = {}
    with pytest.warns(UserWarning, match="schedule buffers"):
        m = load_checkpoint(cfg, {"state_dict": no_sched}, device="cpu", report=rep)
    assert rep["missing"] == [] and len(rep["missing_schedule_buffers"]) == len(sd) - len(no_sched)
    ref = SVCModel(cfg)
    assert torch.equal(m.model.diffusion.naive_noise_predictor.posterior_mean_coef1, ref.model.diffusion.naive_noise_predictor.posterior_mean_coef1)
    no_spec = {k: v for k, v in sd.items() if not k.endswith("spec_min")}
    with pytest.raises(KeyError, match="spec_min"):
        load_checkpoint(cfg, {"state_dict": no_spec}, device="cpu")
    # non-allowlisted object beside the weights
    torch.save({"state_dict": sd, "callbacks": {"x": _Opaque()}}, tmp_path / "opaque.ckpt")
    with pytest.raises(Exception):
        load_checkpoint(cfg, str(tmp_path / "opaque.ckpt"), device="cpu", weights_only=True)
    with pytest.warns(UserWarning, match="full unpickler"):
        m2 = load_checkpoint(cfg, str(tmp_path / "opaque.ckpt"), device="cpu", report=rep)
    assert rep["missing"] == [] and rep["missing_schedule_buffers"] == []
    assert torch.equal(m2.model.text_encoder.projection.weight, sd["model.text_encoder.projection.weight"])
