"""The objects the reference's caller holds, for inference on MI355X:

* `SVCModel(config)` -- what `tools/diffusion/inference.py` keeps as `self.model`: the inference-relevant part of
  `DiffSingerLightning.__init__` (fish_diffusion/archs/diffsinger/diffsinger.py:184-214): `.model` (DiffSinger), `.ema_model` when
  the config carries `ema_momentum`, `.vocoder` (built through VOCODERS, frozen), `.config`.  No optimisers, no LoRA, no logging:
  training is out of scope.
* `load_checkpoint(config, checkpoint, device, model_cls)` -- `fish_diffusion/utils/inference.py:6-32`: `torch.load`, unwrap the Lightning
  `"state_dict"`, drop `vocoder.*`, `load_state_dict(strict=False)`, `.to(device)`, `.eval()` -- PLUS the coverage assertion SURVEY
  section 7 asked for: the reference's `strict=False` lets a checkpoint with a missing `denoise_fn.*` / encoder key load silently onto
  random initialisation.  Here every parameter and every data-carrying buffer (`spec_min` / `spec_max`, `positional_embedding`) of `model.*`
  (and of `ema_model.*` when the config builds one) must have come from the checkpoint, otherwise `KeyError` names what is missing
  (`allow_missing=True` restores the reference's behaviour); schedule buffers the constructor derives from the config only warn.
* `inference_model(m)` -- the `ema_model` preference of `SVCInference.forward` (tools/diffusion/inference.py:134-138).
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
from torch import nn

from .diffsinger import DiffSinger, _cfg_get
from .registry import VOCODERS


def _plain(cfg) -> dict:
    """A dict copy of an mmengine ConfigDict / dict node (the registries' build() takes plain dicts)."""
    if hasattr(cfg, "to_dict"):
        return dict(cfg.to_dict())
    return dict(cfg)


class SVCModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        model_cfg = _cfg_get(config, "model")
        if model_cfg is None:
            raise KeyError("config.model is missing")
        mtype = _cfg_get(model_cfg, "type")
        if mtype == "GradTTS":
            raise NotImplementedError("GradTTS is not part of this path (SURVEY section 2): DiffSinger only")
        if _cfg_get(config, "lora"):
            raise NotImplementedError("LoRA checkpoints (diffsinger.py:190-209) are a training feature and are not loaded here")
        self.config = config
        self.model = DiffSinger(model_cfg)
        self.ema_momentum = _cfg_get(config, "ema_momentum")
        if self.ema_momentum is not None:            # diffsinger.py:196-206: a second copy whose weights the checkpoint carries as ema_model.*
            self.ema_model = DiffSinger(model_cfg)
            self.ema_model.load_state_dict(self.model.state_dict())
            self.ema_model.eval()
            for p in self.ema_model.parameters():
                p.requires_grad_(False)
        voc_cfg = _cfg_get(model_cfg, "vocoder")
        if voc_cfg is None:
            raise KeyError("config.model.vocoder is missing (diffsinger.py:212)")
        self.vocoder = VOCODERS.build(_plain(voc_cfg))
        self.vocoder.freeze()

    @property
    def device(self):
        return next(self.parameters()).device


def inference_model(m: nn.Module) -> nn.Module:
    """`self.model.ema_model if hasattr(self.model, "ema_model") else self.model.model` (tools/diffusion/inference.py:134-138)."""
    return m.ema_model if hasattr(m, "ema_model") else m.model


# Buffers `__init__` recomputes from the config alone (diffusion.py:81-90, noise_predictor.py:29-71,115): a reference checkpoint saved without
# them loads and runs correctly under the reference's strict=False, so their absence is reported, not fatal.  Everything else -- every parameter,
# and the buffers that carry data (`spec_min` / `spec_max`, `positional_embedding`) -- must come from the checkpoint.
_SCHEDULE_BUFFERS = ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod")


def _is_schedule_buffer(key: str) -> bool:
    if ".diffusion." not in "." + key:
        return False
    tail = key.split("diffusion.", 1)[1]
    return tail in _SCHEDULE_BUFFERS or tail.startswith(("naive_noise_predictor.", "plms_noise_predictor."))


def _uncovered(model: nn.Module, loaded: Iterable[str]) -> list:
    have = set(loaded)
    return sorted(k for k in model.state_dict() if not k.startswith("vocoder.") and k not in have)


def _torch_load(path: str, weights_only: Optional[bool]):
    """The reference's environment (torch < 2.6) unpickles whatever a Lightning `.ckpt` holds; torch >= 2.6 defaults to `weights_only=True`
    and refuses optimizer / callback state with non-allowlisted classes.  A checkpoint is code, so the full unpickler is OPT-IN:
    `None` / `True` use the safe loader only (a refusal is re-raised with a hint); `weights_only=False`, or the environment variable
    `FISHDX_UNSAFE_LOAD=1` with `None`, unpickle as the reference's `torch.load` does.  Only `pickle.UnpicklingError` is treated as a
    refusal -- I/O errors and corrupt files propagate as they are."""
    import pickle
    if weights_only is None and os.environ.get("FISHDX_UNSAFE_LOAD", "") == "1":
        weights_only = False
    if weights_only is False:
        return torch.load(path, map_location="cpu", weights_only=False)
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        raise pickle.UnpicklingError(
            f"{path}: the weights-only loader refused this checkpoint ({str(e).splitlines()[0][:160]}).  If you trust the file, pass "
            "weights_only=False (or set FISHDX_UNSAFE_LOAD=1) to unpickle it in full, as the reference's torch.load does") from e


def load_checkpoint(config, checkpoint, device="cuda", model_cls=SVCModel, allow_missing: bool = False, report: Optional[dict] = None,
                    weights_only: Optional[bool] = None):
    """utils/inference.py:6-32 with key-coverage checking.  `checkpoint`: a path (file, or a directory whose naturally-sorted last
    entry is taken, tools/diffusion/inference.py:67-74) or an already loaded dict.  `report`, if given, receives
    {"missing": [...], "missing_schedule_buffers": [...], "unexpected": [...], "loaded": n}.  Missing PARAMETERS or data buffers raise
    `KeyError` unless `allow_missing`; missing config-derived schedule buffers (which `__init__` has already computed from the config)
    only warn -- the sampler then runs on the config's schedule, as the reference would."""
    model = model_cls(config)
    if isinstance(checkpoint, (str, os.PathLike)):
        path = os.fspath(checkpoint)
        if os.path.isdir(path):
            import re
            names = sorted(os.listdir(path), key=lambda s: [int(t) if t.isdigit() else t.lower() for t in re.split(r"(\d+)", s)])
            if not names:
                raise FileNotFoundError(f"no checkpoints under {path}")
            path = os.path.join(path, names[-1])
        state_dict = _torch_load(path, weights_only)
    else:
        state_dict = checkpoint
    if "state_dict" in state_dict:          # saved by Lightning
        state_dict = state_dict["state_dict"]
    state_dict = {k: v for k, v in state_dict.items() if not k.startswith("vocoder.")}
    result = model.load_state_dict(state_dict, strict=False)
    uncovered = _uncovered(model, state_dict.keys())
    sched = [k for k in uncovered if _is_schedule_buffer(k)]
    missing = [k for k in uncovered if not _is_schedule_buffer(k)]
    if report is not None:
        report.update(missing=missing, missing_schedule_buffers=sched, unexpected=sorted(result.unexpected_keys),
                      loaded=len(state_dict) - len(result.unexpected_keys))
    if sched:
        import warnings
        warnings.warn(f"checkpoint carries no schedule buffers ({sched[0]}, ... {len(sched)} keys): using the ones computed from the config")
    if missing and not allow_missing:
        shown = ", ".join(missing[:8]) + (f", ... ({len(missing)} keys)" if len(missing) > 8 else "")
        raise KeyError("checkpoint does not cover the model: " + shown + " -- the reference loads with strict=False and would leave these "
                       "at their constructor values (random initialisation for weights); pass allow_missing=True for that behaviour")
    model.to(device)
    model.eval()
    return model
