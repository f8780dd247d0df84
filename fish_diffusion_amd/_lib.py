"""ctypes binding of libfishdx.so (include/fishdx.h).  No fallback: if the library or a gfx950 device
is missing, everything here raises."""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDX_LIB_PATH") or os.path.join(_HERE, "csrc", "libfishdx.so")  # env override: trace build

FDX_ROW = 16
SAMPLER_NAIVE, SAMPLER_UNIPC, SAMPLER_PLMS = 0, 1, 2
MEL_LINEAR, MEL_LN, MEL_LOG10 = 0, 1, 2
PROF_WN_CONVGATE, PROF_WN_OUTPROJ, PROF_NSF_RESBLOCK, PROF_RG_RESBLOCK, PROF_CN_PWCONV1, PROF_TD_ATTN = 0, 1, 2, 3, 4, 5
MAX_STAGES, MAX_RESK, MAX_DIL = 8, 4, 4


class WavenetDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("mel_channels", "d_encoder", "residual_channels", "residual_layers",
                                       "dilation_cycle", "use_linear_bias")]


class ConvNextDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("mel_channels", "dim", "mlp_factor", "condition_dim", "num_layers", "dilation_cycle",
                                       "cross_attention")]


class TfdecDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("mel_channels", "dim", "mlp_factor", "condition_dim", "num_layers", "n_positions")]


class NsfDesc(C.Structure):
    _fields_ = [("num_mels", C.c_int), ("upsample_initial_channel", C.c_int), ("n_stages", C.c_int),
                ("upsample_rates", C.c_int * MAX_STAGES), ("upsample_kernel_sizes", C.c_int * MAX_STAGES),
                ("n_resblock_kernels", C.c_int), ("resblock_kernel_sizes", C.c_int * MAX_RESK),
                ("n_dilations", C.c_int), ("resblock_dilations", (C.c_int * MAX_DIL) * MAX_RESK),
                ("resblock_type", C.c_int), ("sampling_rate", C.c_int), ("hop_size", C.c_int),
                ("harmonic_num", C.c_int)]


class RefineGanDesc(C.Structure):
    _fields_ = [("sampling_rate", C.c_int), ("hop_length", C.c_int), ("n_down", C.c_int), ("downsample_rates", C.c_int * MAX_STAGES),
                ("n_up", C.c_int), ("upsample_rates", C.c_int * MAX_STAGES), ("num_mels", C.c_int), ("start_channels", C.c_int),
                ("leaky_relu_slope", C.c_float), ("template_sine", C.c_int)]


class FeatureTerm(C.Structure):
    _fields_ = [("kind", C.c_int), ("per_frame", C.c_int), ("preproc", C.c_int), ("src_frames", C.c_int), ("values", C.c_void_p),
                ("w", C.c_void_p), ("b", C.c_void_p), ("p0", C.c_float), ("p1", C.c_float),
                ("neck", C.c_int), ("neck_w", C.c_void_p), ("neck_b", C.c_void_p)]


TERM_VECTOR, TERM_EMBEDDING, TERM_SCALAR_LINEAR = 0, 1, 2
PRE_NONE, PRE_PITCH_TO_SCALE = 0, 1
MAX_FEATURE_TERMS = 6
MAX_NECK = 32
ACT_NONE, ACT_SILU = 0, 1


class MelDesc(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("n_fft", C.c_int), ("win_size", C.c_int), ("hop", C.c_int),
                ("n_mels", C.c_int), ("f_min", C.c_float), ("f_max", C.c_float)]


_P = C.c_void_p
_SIGS = {
    "fdx_version": (C.c_int, []),
    "fdx_device_available": (C.c_int, []),
    "fdx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "fdx_destroy": (C.c_int, [_P]),
    "fdx_last_error": (C.c_char_p, [_P]),
    "fdx_wavenet_num_weights": (C.c_int, [C.POINTER(WavenetDesc)]),
    "fdx_wavenet_packed_bytes": (C.c_int, [C.POINTER(WavenetDesc), C.POINTER(C.c_size_t)]),
    "fdx_wavenet_pack": (C.c_int, [C.POINTER(WavenetDesc), C.POINTER(_P), C.c_int, _P, C.c_size_t]),
    "fdx_wavenet_attach": (C.c_int, [_P, C.POINTER(WavenetDesc), _P, C.c_size_t]),
    "fdx_wavenet_bf16_packed_bytes": (C.c_int, [C.POINTER(WavenetDesc), C.POINTER(C.c_size_t)]),
    "fdx_wavenet_bf16_pack": (C.c_int, [C.POINTER(WavenetDesc), C.POINTER(_P), C.c_int, _P, C.c_size_t]),
    "fdx_wavenet_bf16_attach": (C.c_int, [_P, _P, C.c_size_t]),
    "fdx_wavenet_f16s_enable": (C.c_int, [_P, C.c_int]),
    "fdx_wavenet_bf16_from_arena": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "fdx_wavenet_prepare": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "fdx_wavenet_forward": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P]),
    "fdx_convnext_num_weights": (C.c_int, [C.POINTER(ConvNextDesc)]),
    "fdx_convnext_packed_bytes": (C.c_int, [C.POINTER(ConvNextDesc), C.POINTER(C.c_size_t)]),
    "fdx_convnext_pack": (C.c_int, [C.POINTER(ConvNextDesc), C.POINTER(_P), C.c_int, _P, C.c_size_t]),
    "fdx_convnext_attach": (C.c_int, [_P, C.POINTER(ConvNextDesc), _P, C.c_size_t]),
    "fdx_convnext_prepare": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "fdx_convnext_forward": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P]),
    "fdx_tfdec_num_weights": (C.c_int, [C.POINTER(TfdecDesc)]),
    "fdx_tfdec_packed_bytes": (C.c_int, [C.POINTER(TfdecDesc), C.POINTER(C.c_size_t)]),
    "fdx_tfdec_pack": (C.c_int, [C.POINTER(TfdecDesc), C.POINTER(_P), C.c_int, _P, C.c_size_t]),
    "fdx_tfdec_attach": (C.c_int, [_P, C.POINTER(TfdecDesc), _P, C.c_size_t]),
    "fdx_tfdec_prepare": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "fdx_tfdec_forward": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P]),
    "fdx_sampler_run": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, C.c_uint64, _P, _P]),
    "fdx_sampler_run_ragged": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, C.c_uint64, _P, _P]),
    "fdx_sampler_set_items": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "fdx_q_sample": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_float, C.c_float, _P, _P, _P]),
    "fdx_denorm_spec": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P]),
    "fdx_randn": (C.c_int, [_P, _P, C.c_size_t, C.c_uint64, C.c_uint64, _P]),
    "fdx_nsf_num_weights": (C.c_int, [C.POINTER(NsfDesc)]),
    "fdx_nsf_packed_bytes": (C.c_int, [C.POINTER(NsfDesc), C.POINTER(C.c_size_t)]),
    "fdx_nsf_pack": (C.c_int, [C.POINTER(NsfDesc), C.POINTER(_P), C.c_int, _P, C.c_size_t]),
    "fdx_nsf_attach": (C.c_int, [_P, C.POINTER(NsfDesc), _P, C.c_size_t]),
    "fdx_nsf_forward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_float, _P, _P, C.c_uint64, _P, _P]),
    "fdx_nsf_source": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_uint64, _P, _P]),
    "fdx_mel_config": (C.c_int, [_P, C.POINTER(MelDesc)]),
    "fdx_mel_num_frames": (C.c_int, [C.POINTER(MelDesc), C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int)]),
    "fdx_mel_filterbank": (C.c_int, [C.POINTER(MelDesc), _P]),
    "fdx_mel_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P]),
    "fdx_mel_stats": (C.c_int, [_P, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_int)]),
    "fdx_bcast_arena": (C.c_int, [_P, C.c_size_t, _P, C.c_int, _P]),
    "fdx_refinegan_num_weights": (C.c_int, [C.POINTER(RefineGanDesc)]),
    "fdx_refinegan_num_noises": (C.c_int, [C.POINTER(RefineGanDesc)]),
    "fdx_refinegan_packed_bytes": (C.c_int, [C.POINTER(RefineGanDesc), C.POINTER(C.c_size_t)]),
    "fdx_refinegan_pack": (C.c_int, [C.POINTER(RefineGanDesc), C.POINTER(_P), C.c_int, _P, C.c_size_t]),
    "fdx_refinegan_attach": (C.c_int, [_P, C.POINTER(RefineGanDesc), _P, C.c_size_t]),
    "fdx_refinegan_forward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_float, C.POINTER(_P), C.c_uint64, _P, _P]),
    "fdx_features_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.POINTER(FeatureTerm), C.c_int, _P, _P]),
    "fdx_features_forward_ex": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.POINTER(FeatureTerm), C.c_int, C.c_int, _P,
                                           C.c_int, _P, _P]),
    "fdx_features_forward_src": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.POINTER(FeatureTerm),
                                            C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "fdx_features_forward_svs": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P,
                                            C.POINTER(FeatureTerm), C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "fdx_repeat_expand": (C.c_int, [_P, _P, C.c_long, C.c_int, C.c_int, _P, _P]),
    "fdx_debug_conv1d": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_int, _P, _P]),
    "fdx_prof_enable": (C.c_int, [_P, C.c_int]),
    "fdx_prof_select": (C.c_int, [_P, C.c_int]),
    "fdx_prof_read": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fdx_prof_label": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "fdx_graph_stats": (C.c_int, [_P, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_int)]),
    "fdx_prof_calibrate": (C.c_int, [_P, _P, C.POINTER(C.c_double)]),
}
EXPORTS = tuple(_SIGS)

_lib = None
_lib_lock = threading.Lock()


def lib():
    """The loaded library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        with _lib_lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with `python -m fish_diffusion_amd._build` "
                        "(hipcc, gfx950).  fish_diffusion_amd has no CPU / PyTorch fallback.")
                l = C.CDLL(LIB_PATH)
                for name, (res, args) in _SIGS.items():
                    fn = getattr(l, name)  # AttributeError if include/fishdx.h and the .so disagree
                    fn.restype, fn.argtypes = res, args
                _lib = l
    return _lib


_EXC = {-1: ValueError, -2: RuntimeError, -3: RuntimeError, -4: NotImplementedError, -5: MemoryError}


def check(rc: int, handle=None):
    if rc == 0:
        return
    msg = lib().fdx_last_error(handle)
    raise _EXC.get(rc, RuntimeError)((msg or b"").decode() or f"libfishdx error {rc}")


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device: torch.device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; fish_diffusion_amd runs on an MI355X (HIP) device only "
                           "-- there is no CPU path")


class Handle:
    """One fdx context per (module, device).  Calls are serialised by `lock` (the ABI is not re-entrant
    per handle; the reference's flask server may call forward from several threads)."""

    def __init__(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError(f"fish_diffusion_amd needs a HIP device, got {device}")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        self.h = C.c_void_p()
        self.lock = threading.RLock()
        check(lib().fdx_create(self.device.index, C.byref(self.h)))

    def graph_stats(self):
        """(captures, launches, cached) of this handle's recorded sampler graphs."""
        a, b, c = C.c_long(), C.c_long(), C.c_int()
        check(lib().fdx_graph_stats(self.h, C.byref(a), C.byref(b), C.byref(c)), self.h)
        return a.value, b.value, c.value

    _shared = {}

    @classmethod
    def shared(cls, device: torch.device) -> "Handle":
        """A per-device handle for stateless helpers (repeat_expand, ...) that own no weights."""
        device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        hnd = cls._shared.get(device.index)
        if hnd is None:
            hnd = cls._shared[device.index] = cls(device)
        return hnd

    def __del__(self):
        try:
            if self.h:
                lib().fdx_destroy(self.h)
                self.h = None
        except Exception:
            pass


def host_ptr_array(tensors: Sequence[torch.Tensor]):
    """(keepalive list, void** array) of contiguous fp32 CPU tensors."""
    keep = [t.detach().to("cpu", torch.float32).contiguous() for t in tensors]
    arr = (_P * len(keep))(*[t.data_ptr() for t in keep])
    return keep, arr


def pack_to_device(desc, tensors: Sequence[torch.Tensor], kind: str, device: torch.device) -> torch.Tensor:
    """Host-side repack (pure C++) + one H2D copy.  Returns the uint8 arena tensor on `device`."""
    l = lib()
    n_fn, bytes_fn, pack_fn = (getattr(l, f"fdx_{kind}_{s}") for s in ("num_weights", "packed_bytes", "pack"))
    n = n_fn(C.byref(desc))
    if n < 0:
        check(n)
    if n != len(tensors):
        raise ValueError(f"{kind}: expected {n} weight tensors, got {len(tensors)}")
    nbytes = C.c_size_t()
    check(bytes_fn(C.byref(desc), C.byref(nbytes)))
    keep, arr = host_ptr_array(tensors)
    host = torch.empty(nbytes.value, dtype=torch.uint8, pin_memory=torch.cuda.is_available())
    check(pack_fn(C.byref(desc), arr, n, C.c_void_p(host.data_ptr()), nbytes))
    del keep
    return host.to(device, non_blocking=False)


def pack_on_host(desc, tensors: Sequence[torch.Tensor], kind: str) -> np.ndarray:
    """Same as pack_to_device but stays on the host (CPU-only tests of the packing logic)."""
    l = lib()
    n_fn, bytes_fn, pack_fn = (getattr(l, f"fdx_{kind}_{s}") for s in ("num_weights", "packed_bytes", "pack"))
    n = n_fn(C.byref(desc))
    if n < 0:
        check(n)
    nbytes = C.c_size_t()
    check(bytes_fn(C.byref(desc), C.byref(nbytes)))
    keep, arr = host_ptr_array(tensors)
    out = np.empty(nbytes.value // 4, dtype=np.float32)
    check(pack_fn(C.byref(desc), arr, len(keep), C.c_void_p(out.ctypes.data), nbytes))
    return out
