"""The drop-in boundary: the reference selects implementations by `type=` strings through mmengine
registries (fish_diffusion/archs/diffsinger/diffusions/builder.py:7-15, modules/vocoders/builder.py:1-3).

* When the reference package (and mmengine) is importable, `install()` registers the MI355X classes in
  the reference's OWN registries -- under their original names with `override=True` (so the unchanged
  configs, e.g. configs/svc_hubert_soft.py, build the HIP path) and always under the opt-in names
  `WaveNetDenoiserMI355X`, `GaussianDiffusionMI355X`, `NsfHifiGANMI355X`.
* Otherwise (this image: no mmengine) the same three registries exist here with the two methods the
  reference uses, `register_module` and `build`.
"""
from __future__ import annotations


class Registry:
    """`register_module(name=, module=, force=)` (direct or decorator form) and `build(cfg)`."""

    def __init__(self, name: str):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, module=None, force=False):
        def _add(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force and self.module_dict[key] is not cls:
                raise KeyError(f"{key} is already registered in {self.name}")
            self.module_dict[key] = cls
            return cls

        return _add(module) if module is not None else _add

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"cfg must be a dict with a `type` key, got {cfg!r}")
        args = dict(cfg)
        typ = args.pop("type")
        cls = typ if isinstance(typ, type) else self.get(typ)
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**args)


DENOISERS = Registry("denoisers")
DIFFUSIONS = Registry("diffusions")
VOCODERS = Registry("vocoders")


def install(override: bool = True) -> bool:
    """Register the HIP classes in the reference's registries if fish_diffusion is importable.
    Returns True when the reference registries were found."""
    from .diffusion import GaussianDiffusion
    from .nsf_hifigan import NsfHifiGAN
    from .convnext import ConvNext
    from .tfdec import TransformerDecoderDenoiser
    from .wavenet import WaveNet

    try:   # needs the reference package (+ mmengine or a stand-in): tests/test_install.py
        from fish_diffusion.archs.diffsinger.diffusions.builder import DENOISERS as R_DEN, DIFFUSIONS as R_DIF
        from fish_diffusion.modules.vocoders.builder import VOCODERS as R_VOC
    except Exception:
        return False
    for reg, base, cls in ((R_DEN, "WaveNetDenoiser", WaveNet), (R_DEN, "ConvNextDenoiser", ConvNext),
                           (R_DEN, "TransformerDecoderDenoiser", TransformerDecoderDenoiser), (R_DIF, "GaussianDiffusion", GaussianDiffusion),
                           (R_VOC, "NsfHifiGAN", NsfHifiGAN)):
        reg.register_module(name=base + "MI355X", module=cls, force=True)
        if override:
            reg.register_module(name=base, module=cls, force=True)
    return True
