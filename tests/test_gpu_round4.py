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
