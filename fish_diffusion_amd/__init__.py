"""fish_diffusion_amd -- the MI355X-native hot path of fish-diffusion's SVC/SVS inference
(WaveNet denoiser + sampler loop + STFT/mel + NSF-HiFiGAN) behind the reference's registry API.
See DESIGN.md / INTEGRATION.md.  HIP only: no CPU, no PyTorch-eager fallback."""
from .registry import DENOISERS, DIFFUSIONS, VOCODERS, install  # noqa: F401
from .wavenet import WaveNet  # noqa: F401
from .convnext import ConvNext  # noqa: F401
from .tfdec import TransformerDecoderDenoiser  # noqa: F401
from .diffusion import GaussianDiffusion  # noqa: F401
from .nsf_hifigan import NsfHifiGAN, Generator  # noqa: F401
from .mel import PitchAdjustableMelSpectrogram  # noqa: F401
from .diffsinger import ENCODERS, DiffSinger, NaiveProjectionEncoder, pitch_to_scale, repeat_expand  # noqa: F401
from .refinegan import RefineGAN, RefineGANGenerator  # noqa: F401
from .hifisinger import HiFiSinger  # noqa: F401
from .inference import SVCModel, inference_model, load_checkpoint  # noqa: F401
from . import segments  # noqa: F401

__all__ = ["DENOISERS", "DIFFUSIONS", "VOCODERS", "install", "WaveNet", "ConvNext", "TransformerDecoderDenoiser", "GaussianDiffusion", "NsfHifiGAN", "Generator",
           "PitchAdjustableMelSpectrogram", "ENCODERS", "DiffSinger", "NaiveProjectionEncoder", "pitch_to_scale", "repeat_expand", "RefineGAN", "RefineGANGenerator", "HiFiSinger", "SVCModel", "load_checkpoint", "inference_model"]
