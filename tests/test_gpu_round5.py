"""GPU tests added in round 5: ResBlock2 against the REAL reference's output (its in-place activation quirk included), and whatever the
round's kernel work needs pinned (each test says what it holds)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import ROOT, abs_err, load, rel_err, sha1_state  # noqa: F401

pytestmark = pytest.mark.gpu

MEL_REL = 1e-3   # north_star: 1e-3 rel fp32 on mel -- max|a-b| / max|b| over the whole tensor (tests/helpers.py rel_err)
WAV_ABS = 1e-4   # north_star: 1e-4 abs on waveform samples


@pytest.fixture(scope="module")
def dev(lib_built):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


# ------------------------------------------------------------------------------------------------ ResBlock2 (VERDICT r4 missing 2)
@pytest.mark.parametrize("tag", ["small", "long"])
def test_generator_resblock2_matches_reference_golden(dev, tag):
    """models.py:119-158.  The fixture is the waveform of the real `Generator(resblock="2")`; `F.leaky_relu(x, .., inplace=True)` (:152) makes
    `xt + x` add the activated x and hands leaky_relu(x) to the next ResBlock2 of the stage (:426-431) -- the HIP path reads its operands
    through the compounded slope instead of rewriting the stage tensor."""
    from fish_diffusion_amd import NsfHifiGAN
    from oracle import nsf_hifigan_ref
    g = load(f"nsf_rb2_{tag}")
    h = json.loads(str(g["config"]))
    gsd = nsf_hifigan_ref.seeded_generator_state(int(g["seed"]), h)
    assert sha1_state(gsd) == str(g["weights_sha1"])
    B, T = g["mel"].shape[0], g["mel"].shape[-1]
    torch.manual_seed(int(g["noise_seed"]))
    rand_ini = torch.rand(B, 9)
    rand_ini[:, 0] = 0
    src_noise = torch.randn(B, T * h["hop_size"], 9)
    assert torch.equal(rand_ini, g["rand_ini"])
    voc = NsfHifiGAN.from_state(h, gsd).to(dev)
    wav = voc.model(g["mel"].to(dev), g["f0"].to(dev), rand_ini=rand_ini.to(dev), src_noise=src_noise.to(dev))
    assert wav.shape == g["wav"].shape
    err = abs_err(wav.cpu(), g["wav"])
    print(f"resblock2 {tag}: wav abs err {err:.3e}")
    assert err < WAV_ABS
