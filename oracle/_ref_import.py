"""Import the REAL fish-diffusion reference modules from /root/reference on CPU.

TEST INFRASTRUCTURE ONLY.  This file is used solely by ``oracle/make_golden.py``
(in the build container, where ``/root/reference`` is mounted) to pin the
restatement in ``oracle/*_ref.py`` against the reference's own code and to
generate the fixtures under ``tests/golden/``.  Nothing under ``tests/ -m gpu``,
``bench.py`` or ``__graft_entry__.smoke()`` may call it: ``/root/reference`` does
not exist on the GPU box.

The reference needs a handful of packages that are not installed here
(mmengine, pytorch_lightning, librosa, loguru ...).  Only tiny parts of them are
touched by the hot path, so we pre-seed ``sys.modules`` with stand-ins:

* ``mmengine.Registry``       -- used at fish_diffusion/archs/diffsinger/diffusions/builder.py:1
* ``loguru.logger``           -- fish_diffusion/utils/pitch_adjustable_mel.py:6
* ``librosa.filters.mel``     -- fish_diffusion/utils/pitch_adjustable_mel.py:5
  (restated in numpy in oracle/mel_ref.py -- librosa 0.9.1 slaney semantics)
* a namespace stub for ``fish_diffusion.archs.diffsinger`` so that its
  ``__init__`` (which pulls loralib / lightning / wandb) is skipped.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FISH_REFERENCE_ROOT", "/root/reference")


class _Registry:
    """15-line stand-in for mmengine.Registry (register_module + build)."""

    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, module=None, force=False):
        if module is not None:
            self._modules[name or module.__name__] = module
            return module

        def deco(cls):
            self._modules[name or cls.__name__] = cls
            return cls

        return deco

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg):
        cfg = dict(cfg)
        return self._modules[cfg.pop("type")](**cfg)


def _stub(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "fish_diffusion"))


_loaded = {}


def _hifisinger_methods():
    """`HiFiSinger.get_mask_from_lengths`, `.forward_features` and `.forward` compiled from the reference file's own source
    (fish_diffusion/archs/hifisinger/core.py:39-141); the module imports the whole encoders package at the top."""
    import ast
    import torch
    path = os.path.join(REFERENCE_ROOT, "fish_diffusion/archs/hifisinger/core.py")
    with open(path) as f:
        tree = ast.parse(f.read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "HiFiSinger")
    names = ("get_mask_from_lengths", "forward_features", "forward")
    funcs = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=funcs, type_ignores=[]), path, "exec"), ns)
    return tuple(ns[n] for n in names)


def _diffsinger_methods():
    """`DiffSinger.get_mask_from_lengths` and `DiffSinger.forward_features` compiled from the reference file's own source
    (fish_diffusion/archs/diffsinger/diffsinger.py:42-134) -- the module itself imports loralib / lightning / wandb /
    matplotlib at the top and cannot be imported here, the two methods only need torch."""
    import ast
    import torch
    path = os.path.join(REFERENCE_ROOT, "fish_diffusion/archs/diffsinger/diffsinger.py")
    with open(path) as f:
        src = f.read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DiffSinger")
    funcs = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("get_mask_from_lengths", "forward_features")]
    mod = ast.Module(body=funcs, type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, path, "exec"), ns)
    return ns["get_mask_from_lengths"], ns["forward_features"]


def load():
    """Return a dict of the reference classes on the hot path."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")

    from . import mel_ref  # numpy restatement of librosa.filters.mel

    if "mmengine" not in sys.modules:
        _stub("mmengine", Registry=_Registry)
    if "loguru" not in sys.modules:
        class _Log:
            def __getattr__(self, _):
                return lambda *a, **k: None
        _stub("loguru", logger=_Log())
    if "librosa" not in sys.modules:
        filt = _stub("librosa.filters", mel=mel_ref.slaney_mel_filterbank)
        _stub("librosa", filters=filt)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    # namespace stubs: skip package __init__ files that import lightning & co.
    for pkg in ("fish_diffusion", "fish_diffusion.archs", "fish_diffusion.archs.diffsinger",
                "fish_diffusion.archs.diffsinger.diffusions", "fish_diffusion.modules",
                "fish_diffusion.modules.vocoders", "fish_diffusion.modules.vocoders.nsf_hifigan",
                "fish_diffusion.modules.encoders", "fish_diffusion.utils"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REFERENCE_ROOT, *pkg.split("."))]
            sys.modules[pkg] = m

    # alternative denoisers registered by builder.py:3-5 are off the hot path (and llama.py
    # drags in transformers): stand-ins keep the registry import working.
    if "fish_diffusion.modules.llama" not in sys.modules:
        _stub("fish_diffusion.modules.llama", LlamaDenoiser=type("LlamaDenoiser", (), {}))
    def by_path(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REFERENCE_ROOT, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    wavenet = importlib.import_module("fish_diffusion.modules.wavenet")
    convnext = importlib.import_module("fish_diffusion.modules.convnext")   # the real module: torch + wavenet.DiffusionEmbedding only
    diffusion = importlib.import_module("fish_diffusion.archs.diffsinger.diffusions.diffusion")
    noise_predictor = importlib.import_module("fish_diffusion.archs.diffsinger.diffusions.noise_predictor")
    uni_pc = importlib.import_module("fish_diffusion.archs.diffsinger.diffusions.uni_pc")
    nsf_models = by_path("fish_diffusion.modules.vocoders.nsf_hifigan.models",
                         "fish_diffusion/modules/vocoders/nsf_hifigan/models.py")
    pam = by_path("fish_diffusion.utils.pitch_adjustable_mel",
                  "fish_diffusion/utils/pitch_adjustable_mel.py")

    enc_builder = by_path("fish_diffusion.modules.encoders.builder", "fish_diffusion/modules/encoders/builder.py")
    naive = by_path("fish_diffusion.modules.encoders.naive_projection", "fish_diffusion/modules/encoders/naive_projection.py")
    pitch = by_path("fish_diffusion.utils.pitch", "fish_diffusion/utils/pitch.py")
    tensor_utils = by_path("fish_diffusion.utils.tensor", "fish_diffusion/utils/tensor.py")

    _loaded.update(
        ENCODERS=enc_builder.ENCODERS,
        NaiveProjectionEncoder=naive.NaiveProjectionEncoder,
        pitch_to_scale=pitch.pitch_to_scale,
        repeat_expand=tensor_utils.repeat_expand,
        diffsinger_methods=_diffsinger_methods,
        hifisinger_methods=_hifisinger_methods,
        WaveNet=wavenet.WaveNet,
        ConvNext=convnext.ConvNext,
        TransformerDecoderDenoiser=convnext.TransformerDecoderDenoiser,
        GaussianDiffusion=diffusion.GaussianDiffusion,
        DENOISERS=diffusion.DENOISERS,
        DIFFUSIONS=diffusion.DIFFUSIONS,
        NoiseScheduleVP=uni_pc.NoiseScheduleVP,
        UniPC=uni_pc.UniPC,
        noise_predictor=noise_predictor,
        Generator=nsf_models.Generator,
        AttrDict=nsf_models.AttrDict,
        PitchAdjustableMelSpectrogram=pam.PitchAdjustableMelSpectrogram,
    )
    return _loaded
