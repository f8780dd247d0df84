"""Algorithmic work of the bench configurations: model tables, FLOP formulas and -- since round 6 -- the algorithmic
BYTES per launch of every config's dominant kernel (`roofline.algorithmic_bytes`)."""
from __future__ import annotations

WN_CFG = dict(mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=20, dilation_cycle=4,
              use_linear_bias=True)  # configs/_base_/archs/diff_svc_v2.py:27-35
NSF_V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 2, 2],
              upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
              resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=128, n_fft=2048, hop_size=512,
              win_size=2048, sampling_rate=44100, fmin=40, fmax=16000)  # tools/nsf_hifigan/config_v1.json
NSF_V1_256 = dict(NSF_V1, upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], hop_size=256)  # config_v1_256.json
CN_CFG = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=20)   # modules/convnext.py:156-166 defaults
TD_CFG = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=12)   # modules/convnext.py:264-272 defaults
RG_HIFISINGER = dict(sampling_rate=44100, hop_length=256, downsample_rates=[2, 2, 8, 8], upsample_rates=[8, 8, 2, 2],
                     leaky_relu_slope=0.2, num_mels=256, start_channels=16)   # configs/_base_/archs/hifi_svc_v2.py:43-52
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 matrix == vector peak; tools/ubench/mfmaclk.hip measures 155.1 on this part
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0
BASELINE_METRIC = "audio-seconds/sec/GPU (100-step denoise + NSF-HiFiGAN, 44.1 kHz)"   # BASELINE.json "metric", verbatim


# ====================================================================================================== algorithmic work
def wavenet_flops_per_frame(c=WN_CFG):
    C_, L, M, E = c["residual_channels"], c["residual_layers"], c["mel_channels"], c["d_encoder"]
    return 2.0 * (M * C_ + L * (3 * C_ * 2 * C_ + E * 2 * C_ + C_ * 2 * C_) + C_ * C_ + C_ * M)


def wavenet_hoisted_flops_per_frame(c=WN_CFG):
    """The step-invariant part of the above: the L conditioner projections (wavenet.py:108), executed once per utterance."""
    return 2.0 * c["residual_layers"] * c["d_encoder"] * 2 * c["residual_channels"]


def nsf_flops_per_sample(h=NSF_V1):
    """2*MAC of every conv in Generator.forward per OUTPUT sample (1.2737 MFLOP for config_v1)."""
    hop = h["hop_size"]
    C0 = h["upsample_initial_channel"]
    total = 2.0 * h["num_mels"] * C0 * 7 / hop
    rate = 1.0 / hop
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, cout = C0 >> i, C0 >> (i + 1)
        total += 2.0 * cin * cout * k * rate          # ConvTranspose1d: k taps per INPUT sample
        rate *= u
        s = int(round(1.0 / rate))                     # remaining upsampling = noise conv stride
        total += 2.0 * cout * (2 * s if s > 1 else 1) * rate
        for kk, dils in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
            n_convs = len(dils) * (2 if h["resblock"] == "1" else 1)
            total += 2.0 * cout * cout * kk * n_convs * rate
    total += 2.0 * cout * 7
    return total


def e2e_flops(frames_total, n_steps, samples_total, n_utt_frames_hoist, h=NSF_V1, denoise=True):
    """(algorithmic, executed) FLOPs of one bench step.  Algorithmic = the reference's op count (every step pays the
    conditioner projections).  Executed = what the device ran: the conditioner projections once per utterance."""
    voc = nsf_flops_per_sample(h) * samples_total
    if not denoise:
        return voc, voc
    alg = wavenet_flops_per_frame() * frames_total * n_steps + voc
    return alg, alg - wavenet_hoisted_flops_per_frame() * n_utt_frames_hoist * (n_steps - 1)


def refinegan_flops(T, cfg=RG_HIFISINGER):
    """2*MAC of every Conv1d in RefineGANGenerator.forward for ONE item of T frames (refinegan/generator.py:333-423: template_conv,
    the down path's ResBlocks, mel_conv, source_conv, per up stage input_conv + 3 ParallelResBlock branches of 6 convs, output_conv) --
    the quantity torch.utils.flop_counter reports on the reference module (tools/flops_reference.py checks the formula against it)."""
    c, L = cfg["start_channels"], T * cfg["hop_length"]
    fl = 2.0 * c * 7 * L
    length = L
    for r in cfg["downsample_rates"]:
        length //= r
        fl += 2.0 * length * 7 * (2 * c * c + 5 * (2 * c) ** 2)
        c *= 2
    fl += 2.0 * T * 7 * cfg["num_mels"] * c
    c *= 2
    sf0 = 1
    for r in cfg["upsample_rates"][1:]:
        sf0 *= r
    fl += 2.0 * (T * cfg["upsample_rates"][0]) * c * 2 * sf0
    length = T
    for r in cfg["upsample_rates"]:
        length *= r
        n = c // 2
        fl += 2.0 * length * (7 * (c + c // 4) * n + sum(6 * k * n * n for k in (3, 7, 11)))
        c = n
    fl += 2.0 * L * 7 * c
    return fl


def hifisinger_frontend_flops(T, content_dim=768, hidden=256):
    """text Linear + the two feature_fuser Linears (archs/hifisinger/core.py:24-29,70-107); the scalar encoders are O(hidden) per frame."""
    return 2.0 * T * (content_dim * hidden + 2 * hidden * hidden)


def convnext_flops_per_frame(c=CN_CFG):
    """(algorithmic, hoisted) per frame per denoiser call: 2*MAC of every conv / linear of ConvNext.forward (modules/convnext.py:206-262):
    input_projection, conditioner_projection (2 convs), per block condition_projection + depthwise k=7 + pwconv1 + pwconv2, output_projection.
    Hoisted = what the device runs once per utterance instead of once per call (the conditioner MLP and the L condition projections)."""
    M, D, E, L = c["mel_channels"], c["dim"], c["condition_dim"], c["num_layers"]
    H = D * c["mlp_factor"]
    hoist = 2.0 * (E * H + H * D + L * D * D)
    return 2.0 * (M * D + L * (7 * D + 2 * D * H) + D * D + D * M) + hoist, hoist


def tfdec_flops_per_frame(T, c=TD_CFG):
    """(algorithmic, hoisted) per frame per call of TransformerDecoderDenoiser.forward (modules/convnext.py:330-379) at T frames: the 1x1 conv
    projections, per nn.TransformerDecoderLayer the self-attention (in_proj 3 D^2, QK^T + PV = 4 T D, out_proj D^2), the cross-attention
    (q D^2, k / v of the memory 2 D^2, QK^T + PV, out_proj) and the feed-forward (2 D H).  Hoisted: condition_projection (step-invariant)."""
    M, D, E, L = c["mel_channels"], c["dim"], c["condition_dim"], c["num_layers"]
    H = D * c["mlp_factor"]
    hoist = 2.0 * (E * H + H * D)
    gemm = 2.0 * (M * H + H * D + L * (3 * D * D + D * D + D * D + 2 * D * D + D * D + 2 * D * H) + D * D + D * M)
    attn = L * 2 * 4.0 * T * D
    return gemm + attn + hoist, hoist



# ====================================================================================================== algorithmic bytes per launch
# `roofline.algorithmic_bytes`: what ONE launch of the config's dominant kernel must move if every operand is read once and every result
# written once (SURVEY 8d's per-unit figures x the units a launch processes).  fp32 = 4 bytes everywhere.
def convgate_bytes(columns, c=WN_CFG, esz=4):
    """Dilated conv k = 3 + gate (wavenet.py:107-115): weights [2C x 3C] + Y in [C x n] + conditioner slab in [2C x n] (fp32) + Z out [C x n]."""
    C_ = c["residual_channels"]
    return esz * (2 * C_ * 3 * C_ + C_ * columns + C_ * columns) + 4 * 2 * C_ * columns


def resblock_family_bytes(stages, B):
    """Mean bytes per launch over the ResBlock1 convs of `stages` = [(channels, length, kernel sizes, n dilations)]: per ResBlock and dilation
    conv1 reads x / writes xt, conv2 reads xt and the residual x / writes x'; the last conv2 of a ResBlock also folds into the stage's MRF sum
    (reads it, except for the first ResBlock).  Weights once per launch."""
    total, launches = 0.0, 0
    for ch, length, ksizes, nd in stages:
        act = 4.0 * ch * length * B
        for j, k in enumerate(ksizes):
            total += nd * (2 * act) + nd * (3 * act) + (act if j else 0.0) + 2 * nd * 4.0 * ch * ch * k
            launches += 2 * nd
    return total / max(1, launches)


def nsf_resblock_bytes(T, B, h=NSF_V1, min_channels=64):
    """The NSF-HiFiGAN stages whose ResBlock convs run on the timed instantiation (>= 64 channels; models.py:103-110,426-432)."""
    stages, length, C0 = [], T, h["upsample_initial_channel"]
    for i, u in enumerate(h["upsample_rates"]):
        length *= u
        ch = C0 >> (i + 1)
        if ch >= min_channels:
            stages.append((ch, length, h["resblock_kernel_sizes"], len(h["resblock_dilation_sizes"][0])))
    return resblock_family_bytes(stages, B)


def refinegan_resblock_bytes(T, B, cfg=RG_HIFISINGER, min_channels=64):
    """RefineGAN's ResBlock convs on the same instantiation: the up path's ParallelResBlocks (k = 3 / 7 / 11, three dilations each) and the
    down path's k = 7 ResBlocks, stages with >= 64 channels (refinegan/generator.py:333-423)."""
    stages, c, length = [], cfg["start_channels"], T * cfg["hop_length"]
    for r in cfg["downsample_rates"]:
        length //= r
        c *= 2
        if c >= min_channels:
            stages.append((c, length, [7], 3))
    c *= 2
    length = T
    for r in cfg["upsample_rates"]:
        length *= r
        c //= 2
        if c >= min_channels:
            stages.append((c, length, [3, 7, 11], 3))
    return resblock_family_bytes(stages, B)


def pwconv1_bytes(columns, c=CN_CFG):
    """ConvNext pwconv1 with the LayerNorm folded in (convnext.py:80-82): weights [H x D] + u in [D x n] + group statistics [n x 32] + hidden out [H x n]."""
    D = c["dim"]
    H = D * c["mlp_factor"]
    return 4 * (H * D + D * columns + 32 * columns + H * columns)


def attention_bytes(T, B, c=TD_CFG):
    """One attention launch of the decoder layer (8 heads): Q, K, V read, O written, [D x T] each."""
    return 4 * 4 * c["dim"] * T * B
