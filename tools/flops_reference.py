#!/usr/bin/env python3
"""BUILD-BOX ONLY (needs /root/reference): bench.py's algorithmic FLOP formulas for the SURVEY 8(f) rows checked against
`torch.utils.flop_counter.FlopCounterMode` run on the REAL reference modules (the procedure BASELINE.md section 2 used for the WaveNet and
NSF-HiFiGAN figures): RefineGANGenerator at the svc_hifisinger_v2 geometry (num_mels = 256), ConvNext, TransformerDecoderDenoiser.

    python tools/flops_reference.py > profiles/r04_flops_reference_check.json

flop_counter counts 2*MAC of conv / linear / (b)mm / sdpa.  The one thing it cannot see is nn.MultiheadAttention's fused inference fast
path (`_native_multi_head_attention`), so that path is disabled for the count (same arithmetic through linear + sdpa).  T is kept small (the counts are exactly
linear in T except the attention term, which the formula carries as 4 T^2 D per attention)."""
import importlib.util
import json
import os
import sys

import torch
from torch.utils.flop_counter import FlopCounterMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import _ref_import  # noqa: E402

R = _ref_import.load()
out = {"what": "bench.py's algorithmic FLOP formulas vs torch.utils.flop_counter on the real reference modules (CPU, build box)", "rows": {}}


def count(fn):
    with FlopCounterMode(display=False) as fc:
        with torch.no_grad():
            fn()
    return float(fc.get_total_flops())


# ---- RefineGANGenerator, svc_hifisinger_v2 geometry
spec = importlib.util.spec_from_file_location("fish_diffusion_refinegan_generator",
                                              os.path.join(_ref_import.REFERENCE_ROOT, "fish_diffusion/modules/vocoders/refinegan/generator.py"))
rg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rg)
cfg = dict(bench.RG_HIFISINGER)
gen = rg.RefineGANGenerator(**cfg).eval()
for T in (24, 40):
    mel = torch.randn(1, cfg["num_mels"], T)
    f0 = torch.full((1, 1, T), 220.0)
    got = count(lambda: gen(mel, f0))
    want = bench.refinegan_flops(T, cfg)
    out["rows"][f"refinegan_hifisinger_v2_T{T}"] = {"flop_counter": got, "formula": want, "ratio": got / want}
out["rows"]["refinegan_hifisinger_v2_T1722_formula_GFLOP"] = bench.refinegan_flops(1722, cfg) / 1e9

# ---- ConvNext (per frame per call)
T = 64
cn = R["ConvNext"](**bench.CN_CFG).eval()
x, t, c = torch.randn(1, 128, T), torch.tensor([10]), torch.randn(1, 256, T)
got = count(lambda: cn(x, t, c))
per_frame, hoist = bench.convnext_flops_per_frame()
D, H, L = 512, 2048, 20
t_indep = 2.0 * (D * H + H * D + L * D * D)   # the step-embedding MLP and the L diffusion_step_projections act on ONE column per call: not per frame
out["rows"]["convnext_T64"] = {"flop_counter": got, "formula": per_frame * T + t_indep, "ratio": got / (per_frame * T + t_indep),
                               "per_frame_MFLOP": per_frame / 1e6, "hoisted_per_frame_MFLOP": hoist / 1e6,
                               "note": "formula = per-frame part x T + the T-independent step-embedding part (0.03 % at T = 861, not carried in bench.py)"}

# ---- TransformerDecoderDenoiser
td = R["TransformerDecoderDenoiser"](**bench.TD_CFG).eval()
# nn.MultiheadAttention's fused inference fast path (taken by the SELF-attention: query is key is value, eval mode, no grad) is one
# opaque aten op flop_counter does not price; with it disabled the same arithmetic runs as linear + sdpa, which it does
torch.backends.mha.set_fastpath_enabled(False)
for T in (32, 64):
    x, t, c = torch.randn(1, 128, T), torch.tensor([10]), torch.randn(1, 256, T)
    got = count(lambda: td(x, t, c))
    per_frame, hoist = bench.tfdec_flops_per_frame(T)
    t_indep = 2.0 * (D * H + H * D)
    attn = 12 * 2 * 4.0 * T * 512 * T     # QK^T + PV of the 24 attentions: CPU sdpa (`_scaled_dot_product_flash_attention_for_cpu`) is not priced by flop_counter
    out["rows"][f"tfdec_T{T}"] = {"flop_counter": got, "formula": per_frame * T + t_indep, "formula_without_attention_products": per_frame * T + t_indep - attn,
                                  "ratio_without_attention_products": got / (per_frame * T + t_indep - attn), "per_frame_MFLOP_at_this_T": per_frame / 1e6,
                                  "note": "flop_counter prices every linear of the decoder layers but not the CPU sdpa kernel; the formula adds 4 T^2 D per attention"}
json.dump(out, sys.stdout, indent=1)
print()
