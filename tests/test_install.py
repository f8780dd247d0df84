"""CPU: `fish_diffusion_amd.install()` against the reference's REAL registries (archs/diffsinger/diffusions/builder.py,
modules/vocoders/builder.py imported unmodified through oracle/_ref_import.py's mmengine stand-in): the north_star claim "drops in
under the existing configs" -- the `model.diffusion` dict of configs/_base_/archs/diff_svc_v2.py, built by the reference's own
DIFFUSIONS registry, yields the MI355X classes.  Needs /root/reference (present in the build container, absent on the GPU box)."""
import importlib
import runpy
import os

import pytest

from oracle import _ref_import

pytestmark = pytest.mark.skipif(not _ref_import.available(), reason="reference tree not present (GPU box)")


def test_install_registers_into_the_reference_registries_and_builds_the_shipped_config(lib_built):
    import fish_diffusion_amd
    from fish_diffusion_amd import registry
    R = _ref_import.load()
    builder = importlib.import_module("fish_diffusion.archs.diffsinger.diffusions.builder")
    voc_builder = importlib.import_module("fish_diffusion.modules.vocoders.builder")
    assert builder.DENOISERS is R["DENOISERS"] and builder.DIFFUSIONS is R["DIFFUSIONS"]
    regs = (builder.DENOISERS, builder.DIFFUSIONS, voc_builder.VOCODERS)
    saved = [dict(r._modules) for r in regs]
    ref_wavenet, ref_diffusion = builder.DENOISERS.get("WaveNetDenoiser"), builder.DIFFUSIONS.get("GaussianDiffusion")
    assert ref_wavenet is R["WaveNet"] and ref_diffusion is R["GaussianDiffusion"]
    try:
        # opt-in names only: the reference's own classes stay where they are
        assert registry.install(override=False) is True
        assert builder.DENOISERS.get("WaveNetDenoiser") is ref_wavenet
        assert builder.DENOISERS.get("WaveNetDenoiserMI355X") is fish_diffusion_amd.WaveNet
        assert builder.DIFFUSIONS.get("GaussianDiffusionMI355X") is fish_diffusion_amd.GaussianDiffusion
        assert voc_builder.VOCODERS.get("NsfHifiGANMI355X") is fish_diffusion_amd.NsfHifiGAN
        # default: the unchanged configs build the HIP path
        assert fish_diffusion_amd.install() is True
        cfg = runpy.run_path(os.path.join(_ref_import.REFERENCE_ROOT, "configs", "_base_", "archs", "diff_svc_v2.py"))["model"]
        diff = builder.DIFFUSIONS.build(cfg["diffusion"])                       # the reference's registry, the reference's config dict
        assert type(diff) is fish_diffusion_amd.GaussianDiffusion
        assert type(diff.denoise_fn) is fish_diffusion_amd.WaveNet
        assert (diff.denoise_fn.residual_channels, diff.denoise_fn.n_layers, diff.denoise_fn.dilation_cycle) == (512, 20, 4)
        assert diff.sampler_interval == 10 and diff.noise_predictor == "unipc" and diff.mel_bins == 128
        # the checkpoint contract: the reference module's state dict loads key for key
        ref_net = ref_wavenet(**{k: v for k, v in cfg["diffusion"]["denoiser"].items() if k != "type"})
        assert set(ref_net.state_dict()) == set(diff.denoise_fn.state_dict())
        diff.denoise_fn.load_state_dict(ref_net.state_dict(), strict=True)
        den = builder.DENOISERS.build(dict(type="ConvNextDenoiser", mel_channels=128, dim=64, mlp_factor=2, condition_dim=256, num_layers=2))
        assert type(den) is fish_diffusion_amd.ConvNext
        # the vocoder entry validates its kwargs against the json like the reference (nsf_hifigan.py:64-70): building needs a checkpoint
        # file, so only the lookup is checked here
        assert voc_builder.VOCODERS.get("NsfHifiGAN") is fish_diffusion_amd.NsfHifiGAN
    finally:
        for r, s in zip(regs, saved):
            r._modules.clear()
            r._modules.update(s)
    assert builder.DENOISERS.get("WaveNetDenoiser") is ref_wavenet
