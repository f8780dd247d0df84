"""Shared test helpers: golden loading, seeded synthetic inputs, error metrics, and a numpy emulation of the
MFMA conv kernel's data path (lane-level index math of fish_diffusion_amd/csrc/convgemm.hip.h) used to check
the host-side packing without a GPU."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

WN_SMALL = dict(mel_channels=128, d_encoder=256, residual_channels=64, residual_layers=4, dilation_cycle=4, use_linear_bias=True)
WN_FULL = dict(mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=20, dilation_cycle=4, use_linear_bias=True)
CN_SMALL = dict(mel_channels=128, dim=64, mlp_factor=2, condition_dim=256, num_layers=4, dilation_cycle=4)
CN_FULL = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=20, dilation_cycle=4)
TD_SMALL = dict(mel_channels=128, dim=128, mlp_factor=2, condition_dim=256, num_layers=2)
TD_FULL = dict(mel_channels=128, dim=512, mlp_factor=4, condition_dim=256, num_layers=12)


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fb" and z[k].ndim > 0 else z[k]) for k in z.files}


def sha1_state(sd):
    h = hashlib.sha1()
    for k in sorted(sd):
        h.update(np.ascontiguousarray(sd[k].detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b| -- the 'rel fp32' figure of north_star (global, not per element)."""
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def abs_err(a, b) -> float:
    return float((a.double() - b.double()).abs().max())


def wavenet_sd(cfg, seed):
    from oracle import wavenet_ref
    kw = {k: v for k, v in cfg.items() if k != "dilation_cycle"}
    return wavenet_ref.seeded_wavenet_state(seed, **kw)


def convnext_sd(cfg, seed):
    from oracle import convnext_ref
    return convnext_ref.seeded_state(seed, **{k: v for k, v in cfg.items() if k != "dilation_cycle"})


def convnext_den(sd, cfg):
    from oracle import convnext_ref
    return lambda x, t, c, xm, cm: convnext_ref.convnext_forward(sd, x, t, c, xm, cm, num_layers=cfg["num_layers"],
                                                                 dilation_cycle=cfg["dilation_cycle"])


def tfdec_sd(cfg, seed):
    from oracle import tfdec_ref
    return tfdec_ref.seeded_state(seed, **cfg)


def tfdec_den(sd, cfg):
    from oracle import tfdec_ref
    return lambda x, t, c, xm, cm: tfdec_ref.tfdec_forward(sd, x, t, c, xm, cm, num_layers=cfg["num_layers"])


def synth_f0(T, frame_rate=44100 / 512):
    t = torch.arange(T, dtype=torch.float32) / frame_rate
    f0 = 220.0 * torch.pow(2.0, 0.3 * torch.sin(2 * np.pi * 0.7 * t))
    a, b = (100, 130) if T > 160 else (T // 3, T // 3 + max(2, T // 8))
    f0[a:b] = 0.0
    return f0


# ---------------------------------------------------------------------------------------------------------
# numpy emulation of convgemm_kernel (same tile / wave / lane decomposition, documented MFMA 32x32x2 layouts)
# ---------------------------------------------------------------------------------------------------------
def _mfma_32x32x2(a, b, acc):
    """a, b: [64] per-lane operands; acc: [16, 64] (reg, lane).  A[i][k] = a[k*32+i], B[k][j] = b[k*32+j]."""
    A = a.reshape(2, 32).T.astype(np.float64)
    Bm = b.reshape(2, 32).astype(np.float64)
    D = A @ Bm  # [32 rows, 32 cols]
    for r in range(16):
        for half in range(2):
            row = (r & 3) + 8 * (r >> 2) + 4 * half
            acc[r, half * 32:(half + 1) * 32] += D[row]
    return acc


def emulate_convgemm(packed, X, *, n_mtiles, RB, cin8, taps, shift0, dshift, T, splitk=True, halo=32):
    """packed: flat float32 [n_mtiles][cin8*taps][RB][64][4]; X: [C_pad, ld] padded rows (valid data at +halo).
    Returns dict (mt, rb) -> [32, ncols] accumulators laid out as logical (row_in_block, col)."""
    n_it = cin8 * taps
    P = packed.reshape(n_mtiles, n_it, RB, 64, 4)
    lanes = np.arange(64)
    half, li = lanes >> 5, lanes & 31
    cols = ((T + 63) // 64) * 64
    out = {}
    for mt in range(n_mtiles):
        tiles = [[np.zeros((32, cols)) for _ in range(RB)]]
        for t0 in range(0, cols, 64):
            acc = np.zeros((RB, 2, 16, 64))
            for it in range(n_it):  # split-K only changes the summation order
                cb, tap = divmod(it, taps)
                for j in range(4):
                    ch = cb * 8 + half * 4 + j
                    for nb in range(2):
                        col = halo + t0 + 2 * li + nb + shift0 + tap * dshift   # lane li owns the column pair (2li, 2li+1)
                        b = X[ch, col]
                        for rb in range(RB):
                            a = P[mt, it, rb, :, j]
                            _mfma_32x32x2(a, b, acc[rb, nb])
            for rb in range(RB):
                for nb in range(2):
                    for r in range(16):
                        for h in range(2):
                            row = (r & 3) + 8 * (r >> 2) + 4 * h
                            tiles[0][rb][row, t0 + nb:t0 + 64:2] = acc[rb, nb, r, h * 32:(h + 1) * 32]
        for rb in range(RB):
            out[(mt, rb)] = tiles[0][rb][:, :T]
    return out


def emulate_convgemm16(packed, X, *, n_mtiles, cin8, taps, shift0, dshift, T, halo=32):
    """numpy emulation of convgemm16_kernel (fish_diffusion_amd/csrc/convgemm16.hip.h): v_mfma_f32_16x16x4_f32 lane maps
    (A: lane l = row l&15, k l>>4;  B: k-row l>>4, column group l&15, value m = column 4*(l&15)+m;  D: row (l>>4)*4+reg).
    packed: flat float32 [n_mtiles][cin8*taps][2][64][4].  Returns dict mt -> [64, T] (tile rows rbk*16 + ...)."""
    n_it = cin8 * taps
    P = packed.reshape(n_mtiles, n_it, 2, 64, 4)
    lanes = np.arange(64)
    lj, lk = lanes & 15, lanes >> 4
    cols = ((T + 63) // 64) * 64
    out = {}
    for mt in range(n_mtiles):
        tile = np.zeros((64, cols))
        for t0 in range(0, cols, 64):
            acc = np.zeros((4, 4, 4, 64))                      # [rbk][m][reg][lane]
            for it in range(n_it):
                cb, tap = divmod(it, taps)
                for h in range(2):
                    ch = cb * 8 + h * 4 + lk                   # per lane: the k-row it loads
                    base = halo + t0 + 4 * lj + shift0 + tap * dshift
                    for m in range(4):
                        b = X[ch, base + m]                    # per-lane B operand of column set m
                        Bm = np.zeros((4, 16))
                        Bm[lk, lj] = b
                        for rbk in range(4):
                            a = P[mt, it, h, :, rbk]           # per-lane A operand of row block rbk
                            Am = np.zeros((16, 4))
                            Am[lj, lk] = a                     # A[i = l&15][k = l>>4]
                            D = Am.astype(np.float64) @ Bm.astype(np.float64)     # [16 rows, 16 col groups]
                            for reg in range(4):
                                acc[rbk, m, reg] += D[lk * 4 + reg, lj]
            for rbk in range(4):
                for m in range(4):
                    for reg in range(4):
                        tile[rbk * 16 + lk * 4 + reg, t0 + 4 * lj + m] = acc[rbk, m, reg]
        out[mt] = tile[:, :T]
    return out
