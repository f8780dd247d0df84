"""CPU: host-side logic of the product -- C-ABI surface, loud failure without a GPU, weight packing (checked by a
numpy emulation of the MFMA kernel's lane-level index math), scalar sampler tables, filterbank, registry."""
import ctypes as C
import json
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import emulate_convgemm16, ROOT, WN_SMALL, emulate_convgemm, wavenet_sd


@pytest.fixture(scope="module")
def lib(lib_built):
    from fish_diffusion_amd import _lib
    return _lib


def test_abi_exports_every_declared_symbol(lib):
    header = open(f"{ROOT}/include/fishdx.h").read()
    declared = set(re.findall(r"\b(fdx_[a-z0-9_]+)\s*\(", header))
    declared -= {"fdx_ctx"}
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    l = lib.lib()
    for name in declared:
        assert hasattr(l, name)
    assert l.fdx_version() >= 100


def test_fails_loudly_without_gpu(lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.lib().fdx_create(0, C.byref(h))
    assert rc == -3 and b"no HIP device" in lib.lib().fdx_last_error(None)
    from fish_diffusion_amd import WaveNet
    net = WaveNet(**WN_SMALL)
    with pytest.raises(RuntimeError, match="no CPU path"):
        net(torch.zeros(1, 128, 8), torch.zeros(1), torch.zeros(1, 256, 8))


def test_bad_configs_raise_like_the_reference(lib):
    from fish_diffusion_amd import DENOISERS, DIFFUSIONS, GaussianDiffusion
    with pytest.raises(ValueError):
        DENOISERS.build(dict(type="WaveNetDenoiser", residual_channels=100))
    with pytest.raises(AssertionError):
        GaussianDiffusion(dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-5], spec_max=None)
    with pytest.raises(AssertionError):
        GaussianDiffusion(dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-5, -4], spec_max=[0, 0])
    d = DIFFUSIONS.build(dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-5], spec_max=[0]))
    assert d.noise_predictor == "unipc" and d.num_timesteps == 1000 and d.mel_bins == 128
    d1 = GaussianDiffusion(dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-5], spec_max=[0], sampler_interval=1)
    assert d1.noise_predictor == "naive"
    with pytest.raises(KeyError):
        DENOISERS.build(dict(type="NoSuchDenoiser"))


def test_state_dict_keys_match_reference_contract(lib):
    from fish_diffusion_amd import GaussianDiffusion, WaveNet
    from oracle import wavenet_ref
    net = WaveNet(**WN_SMALL)
    want = [k for k, _ in wavenet_ref.wavenet_param_shapes(128, 256, 64, 4, True)]
    assert list(net.state_dict().keys()) == want
    sd = wavenet_sd(WN_SMALL, 1)
    assert net.load_state_dict(sd, strict=True)
    d = GaussianDiffusion(dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-5], spec_max=[0])
    keys = set(d.state_dict().keys())
    for k in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "spec_min", "spec_max",
              "naive_noise_predictor.posterior_mean_coef1", "plms_noise_predictor.alphas_cumprod",
              "denoise_fn.residual_layers.3.output_projection.conv.weight"):
        assert k in keys, k
    # no-bias variant (v1 arch)
    assert "mlp.0.linear.bias" not in WaveNet(128, 256, 64, 2, use_linear_bias=False).state_dict()


# ------------------------------------------------------------------ packing, via the numpy kernel emulation
def _wn_offsets(C_, L, M, E):
    """Mirror of wn_layout (fish_diffusion_amd/csrc/wavenet.hip): arena section offsets in floats."""
    def plan(cur, rows, cin, taps, paired, RB=2):
        mt = (rows // 2 + 31) // 32 if paired else (rows + 32 * RB - 1) // (32 * RB)
        w = mt * ((cin + 7) // 8) * taps * RB * 64 * 4
        return dict(w=cur, b=cur + w, mt=mt, cin8=(cin + 7) // 8, taps=taps), cur + w + (rows + 63) // 64 * 64
    cur, out = 0, {}
    for name, rows, cin in (("in_proj", C_, M), ("mlp0", 4 * C_, C_), ("mlp2", C_, 4 * C_), ("dproj", L * C_, C_), ("cond", L * 2 * C_, E)):
        out[name], cur = plan(cur, rows, cin, 1, False, RB=1 if name == "in_proj" else 2)
    for i in range(L):
        out[f"conv{i}"], cur = plan(cur, 2 * C_, C_, 3, True)
        out[f"outp{i}"], cur = plan(cur, 2 * C_, C_, 1, False)
    out["skip_proj"], cur = plan(cur, C_, C_, 1, False, RB=1)
    out["out_proj"], cur = plan(cur, M, C_, 1, False, RB=1)
    out["total"] = cur
    return out


def test_wavenet_packing_matches_kernel_index_math(lib):
    from fish_diffusion_amd import WaveNet
    net = WaveNet(**WN_SMALL)
    sd = wavenet_sd(WN_SMALL, 3)
    net.load_state_dict(sd)
    arena = lib.pack_on_host(net._desc, net._params(), "wavenet")
    off = _wn_offsets(64, 4, 128, 256)
    assert arena.size == off["total"]
    T, halo = 40, 32
    ld = halo + 64 + halo
    g = torch.Generator().manual_seed(0)
    # ---- paired dilated conv of layer 1 (dilation 2): gate rows in rb 0, filter rows in rb 1
    y = torch.randn(64, T, generator=g)
    X = np.zeros((64, ld), np.float32)
    X[:, halo:halo + T] = y.numpy()
    o = off["conv1"]
    # (residual-block weights are packed for the 16x16x4 kernels: tile rows 0..31 = gate, 32..63 = filter)
    acc = emulate_convgemm16(arena[o["w"]:o["b"]], X, n_mtiles=o["mt"], cin8=o["cin8"], taps=3, shift0=-2, dshift=2, T=T)
    ref = F.conv1d(y[None], sd["residual_layers.1.conv_layer.conv.weight"], None, padding=2, dilation=2)[0].numpy()
    for mt in range(o["mt"]):
        np.testing.assert_allclose(acc[mt][:32], ref[mt * 32:mt * 32 + 32], rtol=1e-4, atol=1e-5)          # gate
        np.testing.assert_allclose(acc[mt][32:], ref[64 + mt * 32:64 + mt * 32 + 32], rtol=1e-4, atol=1e-5)  # filter
    # ---- output projection of layer 2: plain 64-row tiles on the 32x32x2 family
    z = torch.randn(64, T, generator=g)
    X = np.zeros((64, ld), np.float32)
    X[:, halo:halo + T] = z.numpy()
    o = off["outp2"]
    acc = emulate_convgemm(arena[o["w"]:o["b"]], X, n_mtiles=o["mt"], RB=2, cin8=o["cin8"], taps=1, shift0=0, dshift=0, T=T)
    ref = (sd["residual_layers.2.output_projection.conv.weight"][:, :, 0] @ z).numpy()
    for mt in range(o["mt"]):
        np.testing.assert_allclose(np.concatenate([acc[(mt, 0)], acc[(mt, 1)]]), ref[mt * 64:mt * 64 + 64], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(arena[o["b"]:o["b"] + 128], sd["residual_layers.2.output_projection.conv.bias"].numpy())
    # ---- concatenated diffusion projection: row l*C + c
    s = torch.randn(64, 5, generator=g)
    X = np.zeros((64, ld), np.float32)
    X[:, halo:halo + 5] = s.numpy()
    o = off["dproj"]
    acc = emulate_convgemm(arena[o["w"]:o["b"]], X, n_mtiles=o["mt"], RB=2, cin8=o["cin8"], taps=1, shift0=0, dshift=0, T=5)
    got = np.concatenate([np.concatenate([acc[(mt, 0)], acc[(mt, 1)]]) for mt in range(o["mt"])])
    ref = torch.cat([sd[f"residual_layers.{i}.diffusion_projection.linear.weight"] @ s for i in range(4)]).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)
    bias = arena[o["b"]:o["b"] + 256]
    np.testing.assert_array_equal(bias, torch.cat([sd[f"residual_layers.{i}.diffusion_projection.linear.bias"] for i in range(4)]).numpy())
    # ---- final projection: 32-row tiles (RB = 1)
    hx = torch.randn(64, T, generator=g)
    X = np.zeros((64, ld), np.float32)
    X[:, halo:halo + T] = hx.numpy()
    o = off["out_proj"]
    acc = emulate_convgemm(arena[o["w"]:o["b"]], X, n_mtiles=o["mt"], RB=1, cin8=o["cin8"], taps=1, shift0=0, dshift=0, T=T)
    got = np.concatenate([acc[(mt, 0)] for mt in range(o["mt"])])
    np.testing.assert_allclose(got, (sd["output_projection.conv.weight"][:, :, 0] @ hx).numpy(), rtol=1e-4, atol=1e-5)
    # ---- conditioner slab bias absorbs the conv bias
    o = off["cond"]
    want = torch.cat([sd[f"residual_layers.{i}.conditioner_projection.conv.bias"] + sd[f"residual_layers.{i}.conv_layer.conv.bias"]
                      for i in range(4)]).numpy()
    np.testing.assert_allclose(arena[o["b"]:o["b"] + 512], want, rtol=0, atol=1e-7)


def test_polyphase_transposed_conv_packing(lib):
    """ups.0 of a tiny generator, emulated, equals F.conv_transpose1d (k=16,u=8,p=4 and k=8,u=2,p=3 geometries)."""
    from fish_diffusion_amd.nsf_hifigan import Generator
    from oracle import nsf_hifigan_ref
    h = dict(nsf_hifigan_ref.CONFIG_V1, upsample_initial_channel=64, upsample_rates=[8, 2], upsample_kernel_sizes=[16, 8],
             hop_size=16, resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1, 3, 5]])
    gen = Generator(h)
    sd = nsf_hifigan_ref.seeded_generator_state(9, h)
    gen.load_folded_state(sd)
    arena = lib.pack_on_host(gen._desc, gen.folded_weights(), "nsf")
    cur = 64 + 64  # src_w, src_b
    cur += 1 * (128 // 8) * 7 * 2 * 64 * 4 + 64  # conv_pre: rows 64 -> 1 m-tile (RB 2), cin8 16, taps 7; bias 64
    T, halo = 12, 32
    g = torch.Generator().manual_seed(1)
    for i, (cin, cout, k, u, taps, dhi) in enumerate([(64, 32, 16, 8, 3, 1), (32, 16, 8, 2, 5, 2)]):
        rows = u * cout
        RB = 1 if rows <= 32 else 2
        mt = (rows + 32 * RB - 1) // (32 * RB)
        wfl = mt * (cin // 8) * taps * RB * 64 * 4
        x = torch.randn(cin, T, generator=g)
        X = np.zeros((cin, halo + 64 + halo), np.float32)
        X[:, halo:halo + T] = x.numpy()
        acc = emulate_convgemm(arena[cur:cur + wfl], X, n_mtiles=mt, RB=RB, cin8=cin // 8, taps=taps, shift0=-dhi, dshift=1, T=T)
        ref = F.conv_transpose1d(x[None], sd[f"ups.{i}.weight"], None, stride=u, padding=(k - u) // 2)[0].numpy()
        got = np.zeros_like(ref)
        for m in range(mt):
            for rb in range(RB):
                blk = acc[(m, rb)]
                for r in range(32):
                    row = m * 32 * RB + rb * 32 + r
                    if row < rows:
                        got[row % cout, (row // cout)::u] = blk[r]
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)
        nk = 2 * 2 if i == 0 else 1
        cur += wfl + (rows + 63) // 64 * 64 + (cout * nk + 63) // 64 * 64 + (cout + 63) // 64 * 64


# ------------------------------------------------------------------ scalar tables
@pytest.mark.parametrize("interval", [50, 10])
def test_unipc_table_matches_oracle_arithmetic(interval):
    """Drive the oracle UniPC with a linear 'denoiser' and replay it from the table: identical x_0."""
    from fish_diffusion_amd import schedule
    from oracle import sampler_ref
    betas = sampler_ref.beta_schedule()
    steps = 1000 // interval
    kind, tab = schedule.sampler_table("unipc", interval=interval)
    assert tab.shape == (steps + 1, 16) and kind == 1
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1, 4, 6, generator=g)
    W = torch.randn(4, 4, generator=g) * 0.3

    def eps_model(x, t):
        return torch.einsum("ij,bjt->bit", W, x) * (1 + 1e-3 * t.view(-1, 1, 1))
    want = sampler_ref.unipc_sample(eps_model, x0, betas, steps)
    # replay with the table (same formulas as the device kernels)
    x = x0.clone()
    def model(xx, row):
        return eps_model(xx, torch.tensor([row[0]]))
    r0 = tab[0]
    m0 = (x - r0[1] * model(x, r0)) / r0[2]
    m1 = None
    for r in range(1, steps + 1):
        t_in, sig, al, c_x, c_m, aB, rk, order, corr, rho0, rho1 = [torch.tensor(v) for v in tab[r, :11]]
        xb = c_x * x - c_m * m0
        xt = xb - aB * (0.5 * ((m1 - m0) / rk)) if order == 2 else xb
        if corr:
            mt = (xt - sig * model(xt, tab[r])) / al
            c = rho0 * ((m1 - m0) / rk) + rho1 * (mt - m0) if order == 2 else rho1 * (mt - m0)
            x = xb - aB * c
            m1, m0 = m0, mt
        else:
            x = xt
    assert torch.equal(x, want)


def test_naive_and_plms_tables():
    from fish_diffusion_amd import schedule
    from oracle import sampler_ref
    betas = sampler_ref.beta_schedule()
    kind, tab = schedule.sampler_table("naive", interval=50, skip_steps=400)
    chunks = schedule.timestep_chunks(1000, 400, 50)
    assert kind == 0 and [int(t) for t in tab[:, 0]] == chunks == list(range(0, 600, 50))[::-1]
    tb = sampler_ref.NaiveTables(betas)
    x, e, n = torch.randn(3), torch.randn(3), torch.randn(3)
    for r, t in enumerate(chunks):
        _, sr, srm1, c1, c2, sc = [torch.tensor(v) for v in tab[r, :6]]
        got = c1 * torch.clamp(sr * x - srm1 * e, -1, 1) + c2 * x + sc * n
        assert torch.equal(got, sampler_ref.naive_step(tb, x, t, e, n))
    kind, tab = schedule.sampler_table("plms", interval=100)
    ac = sampler_ref.f32(np.cumprod(1.0 - betas))
    for r, t in enumerate(schedule.timestep_chunks(1000, 0, 100)):
        _, tp, A, P, Q = [torch.tensor(v) for v in tab[r, :5]]
        assert int(tp) == max(t - 100, 0)
        assert torch.equal(x + A * (P * x - Q * e), sampler_ref.plms_x_pred(ac, x, e, t, int(tp)))
    with pytest.raises(NotImplementedError):
        schedule.sampler_table("euler")
    with pytest.raises(NotImplementedError):
        schedule.make_betas("sigmoid")
    assert np.array_equal(schedule.make_betas("cosine"), sampler_ref.beta_schedule("cosine"))


def test_mel_filterbank_and_frame_count(lib):
    from fish_diffusion_amd import PitchAdjustableMelSpectrogram
    from oracle import mel_ref
    m = PitchAdjustableMelSpectrogram()
    fb = m.filterbank().numpy()
    ref = mel_ref.slaney_mel_filterbank(sr=44100, n_fft=2048, n_mels=128, fmin=40, fmax=16000)
    np.testing.assert_allclose(fb, ref, rtol=2e-6, atol=1e-9)
    for ks, sp, n in ((0, 1.0, 44100), (3, 1.0, 44100), (-5, 1.0, 30000), (12, 1.0, 44100), (0, 1.5, 44100), (0, 1.0, 441000)):
        n_fft, win, hop, pad = mel_ref.stft_geometry(2048, 2048, 512, ks, sp)
        want = 1 + (n + 2 * pad - n_fft) // hop
        assert m.num_frames(n, ks, sp) == want
    with pytest.raises(ValueError):
        m.num_frames(100)


def test_generator_checkpoint_contract(lib, tmp_path):
    """Reference checkpoints are in weight-norm form under `generator.*` or {"generator": sd} (nsf_hifigan.py:38-52)."""
    from fish_diffusion_amd import NsfHifiGAN
    from fish_diffusion_amd.nsf_hifigan import Generator, generator_param_table
    from oracle import nsf_hifigan_ref
    h = dict(nsf_hifigan_ref.CONFIG_V1, upsample_initial_channel=256)
    gen = Generator(h)
    keys = set(gen.state_dict().keys())
    assert "conv_pre.weight_g" in keys and "ups.0.weight_v" in keys and "resblocks.14.convs2.2.weight_g" in keys
    assert "noise_convs.0.weight" in keys and "m_source.l_linear.weight" in keys and "conv_post.weight_g" in keys
    sd = {k: v.clone() for k, v in gen.state_dict().items()}
    (tmp_path / "config.json").write_text(json.dumps(h))
    torch.save({"generator": sd}, tmp_path / "model")
    voc = NsfHifiGAN(str(tmp_path / "model"), sampling_rate=44100, mel_channels=128)
    assert "conv_pre.weight" in voc.model.state_dict() and "conv_pre.weight_g" not in voc.model.state_dict()
    folded = nsf_hifigan_ref.fold_weight_norm(sd)
    for (key, _, _), w in zip(generator_param_table(h), voc.model.folded_weights()):
        assert torch.allclose(w, folded[key], atol=1e-7), key
    torch.save({"state_dict": {"generator." + k: v for k, v in sd.items()}}, tmp_path / "model2")
    NsfHifiGAN(str(tmp_path / "model2"), config_file=str(tmp_path / "config.json"))
    with pytest.raises(ValueError, match="Incorrect value"):
        NsfHifiGAN(str(tmp_path / "model"), sampling_rate=48000)
    voc.freeze()
    assert not any(p.requires_grad for p in voc.parameters())


def test_diffsinger_frontend_state_dict_contract(lib):
    """DiffSinger mirrors the reference's attribute names -> checkpoint keys (diffsinger.py:20-40, naive_projection.py:35-44)."""
    from fish_diffusion_amd import DiffSinger, pitch_to_scale
    cfg = dict(text_encoder=dict(type="NaiveProjectionEncoder", input_size=256, output_size=256),
               speaker_encoder=dict(type="NaiveProjectionEncoder", input_size=10, output_size=256, use_embedding=True),
               pitch_encoder=dict(type="NaiveProjectionEncoder", input_size=1, output_size=256, preprocessing=pitch_to_scale),
               diffusion=dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", mel_channels=128, d_encoder=256,
                                                                      residual_channels=32, residual_layers=2, dilation_cycle=2,
                                                                      use_linear_bias=True), spec_min=[-5], spec_max=[0]))
    m = DiffSinger(cfg)
    keys = set(m.state_dict())
    for k in ("text_encoder.projection.weight", "text_encoder.projection.bias", "speaker_encoder.embedding.weight",
              "pitch_encoder.projection.weight", "pitch_encoder.projection.bias",
              "diffusion.denoise_fn.residual_layers.1.conv_layer.conv.weight", "diffusion.spec_min", "diffusion.betas"):
        assert k in keys, k
    assert tuple(m.pitch_encoder.projection.weight.shape) == (256, 1)
    f0 = torch.tensor([[0.0, 50.0, 575.0, 1100.0, 5000.0]])
    assert torch.allclose(pitch_to_scale(f0)[0, :, 0], torch.tensor([0.0, 0.0, 0.5, 1.0, 1.0]))
    mask = DiffSinger.get_mask_from_lengths(torch.tensor([3, 1]), 4)
    assert mask.tolist() == [[False, False, False, True], [False, True, True, True]]
    with pytest.raises(NotImplementedError):
        m.forward()
    with pytest.raises(RuntimeError):   # CPU tensors: no fallback
        m.forward_features(torch.tensor([1]), torch.zeros(1, 4, 256), torch.tensor([4]), 4, pitches=torch.zeros(1, 4))
    # use_neck: nn.Sequential(Linear(in, neck), Linear(neck, out)) -> keys projection.0.* / projection.1.* (naive_projection.py:37-41)
    from fish_diffusion_amd.diffsinger import NaiveProjectionEncoder
    enc = NaiveProjectionEncoder(64, 96, use_neck=True, neck_size=8)
    assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == {
        "projection.0.weight": (8, 64), "projection.0.bias": (8,), "projection.1.weight": (96, 8), "projection.1.bias": (96,)}
    w, b, neck, nw, nb = enc.linear_params()
    assert neck == 8 and tuple(w.shape) == (96, 8) and tuple(nw.shape) == (8, 64) and float(nb.abs().max()) == 0.0
    assert NaiveProjectionEncoder(10, 96, use_embedding=True, use_neck=True).use_neck is False    # the embedding wins, as in the reference
    with pytest.raises(ValueError):
        NaiveProjectionEncoder(64, 96, use_neck=True, neck_size=64)


# ------------------------------------------------------------------ ConvNext denoiser: packing + arena layout, via the numpy kernel emulation
def _cn_offsets(D, Hf, L, M, E):
    """Mirror of cn_layout (fish_diffusion_amd/csrc/convnext.hip): arena section offsets in floats."""
    H = D * Hf
    r64 = lambda n: (n + 63) // 64 * 64

    def plan(cur, rows, cin, RB=2):
        mt = (rows + 32 * RB - 1) // (32 * RB)
        w = mt * ((cin + 7) // 8) * RB * 64 * 4
        return dict(w=cur, b=cur + w, mt=mt, cin8=(cin + 7) // 8, RB=RB, rows=rows), cur + w + r64(rows)
    cur, out = 0, {}
    for name, rows, cin in (("in_proj", D, M), ("emb1", H, D), ("emb3", D, H), ("cond0", H, E), ("cond2", D, H), ("dsp", L * D, D),
                            ("cproj", L * D, D)):
        out[name], cur = plan(cur, rows, cin, RB=1 if name == "in_proj" else 2)
    for i in range(L):
        out[f"pw1_{i}"], cur = plan(cur, H, D, RB=1)        # 32-row tiles since round 6
        out[f"pw2_{i}"], cur = plan(cur, D, H, RB=1)
        out[f"pw2w_{i}"], cur = plan(cur, D, H, RB=2)
        out[f"dw_w{i}"] = cur; cur += r64(D * 7)
        for nm in ("dw_b", "gamma"):
            out[f"{nm}{i}"] = cur; cur += r64(D)
        out[f"lnR{i}"] = cur; cur += r64(H) * 16
        out[f"lnRs{i}"] = cur; cur += r64(H)               # whole-row sums (PRE_LNP)
    out["out0"], cur = plan(cur, D, D, RB=1)
    out["out2"], cur = plan(cur, M, D, RB=1)
    out["total"] = cur
    return out


def test_convnext_arena_forward_emulation_matches_oracle(lib):
    """Runs the whole ConvNext forward on the CPU *from the packed arena*, with the MFMA kernel's lane-level index math
    (tests/helpers.emulate_convgemm) for every GEMM and numpy for the dwconv + LayerNorm kernel, and compares with the oracle:
    checks fdx_convnext_pack, the arena layout and the hoisting algebra (per-layer slabs) without a GPU."""
    from fish_diffusion_amd import ConvNext
    from oracle import convnext_ref, wavenet_ref
    cfg = dict(mel_channels=16, dim=32, mlp_factor=2, condition_dim=24, num_layers=3, dilation_cycle=2)
    D, H, L, M, E = 32, 64, 3, 16, 24
    sd = convnext_ref.seeded_state(5, **{k: v for k, v in cfg.items() if k != "dilation_cycle"})
    net = ConvNext(**cfg)
    net.load_state_dict(sd, strict=True)
    arena = lib.pack_on_host(net._desc, net._params(), "convnext")
    off = _cn_offsets(D, 2, L, M, E)
    assert arena.size == off["total"]
    T, halo = 21, 32

    def gemm(name, X, bias=True):   # X [cin, n] -> [rows, n] (+ bias)
        o = off[name]
        n = X.shape[1]
        Xp = np.zeros((o["cin8"] * 8, halo + (n + 63) // 64 * 64 + halo), np.float32)
        Xp[:X.shape[0], halo:halo + n] = X
        acc = emulate_convgemm(arena[o["w"]:o["b"]], Xp, n_mtiles=o["mt"], RB=o["RB"], cin8=o["cin8"], taps=1, shift0=0, dshift=0, T=n)
        full = np.concatenate([acc[(mt, rb)] for mt in range(o["mt"]) for rb in range(o["RB"])])[:o["rows"]]
        if not bias:
            return full
        return (full + arena[o["b"]:o["b"] + o["rows"]][:, None].astype(np.float64)).astype(np.float32)

    gelu = lambda a: torch.nn.functional.gelu(torch.from_numpy(a)).numpy()
    g = torch.Generator().manual_seed(1)
    x, cond = torch.randn(1, M, T, generator=g), torch.randn(1, E, T, generator=g)
    t = torch.tensor([123.0])
    mask = torch.zeros(1, T, dtype=torch.bool)
    mask[0, 17:] = True
    keep = (~mask[0]).numpy().astype(np.float32)[None]
    # hoisted paths
    c2 = gemm("cond2", gelu(gemm("cond0", cond[0].numpy()))) * keep
    CP = gemm("cproj", c2)                                                   # [L*D, T]
    emb = wavenet_ref.diffusion_embedding(t, D).numpy().T                     # [D, 1]
    SB = gemm("dsp", gemm("emb3", gelu(gemm("emb1", emb))))                   # [L*D, 1]
    # denoiser call
    X = gelu(gemm("in_proj", x[0].numpy())) * keep
    for i in range(L):
        dil = 2 ** (i % 2)
        v = (X + SB[i * D:(i + 1) * D] + CP[i * D:(i + 1) * D]) * keep
        vp = np.pad(v, ((0, 0), (3 * dil, 3 * dil)))
        w = arena[off[f"dw_w{i}"]:off[f"dw_w{i}"] + D * 7].reshape(D, 7)
        u = sum(w[:, k:k + 1] * vp[:, k * dil:k * dil + T] for k in range(7)) + arena[off[f"dw_b{i}"]:off[f"dw_b{i}"] + D][:, None]
        mu = u.mean(0, keepdims=True)
        var = ((u - mu) ** 2).mean(0, keepdims=True)
        # LayerNorm folded into pwconv1 (PRE_LN): the B operand is centred per group of 32 channels, the epilogue adds the
        # group-offset term R (mean_g - mean) and scales by rstd; the affine part lives inside the packed weights / bias
        o1 = off[f"pw1_{i}"]
        ug = u.reshape(D // 32, 32, T)
        mean_g = ug.mean(1)                                                    # [G, T]
        m2_g = ((ug - mean_g[:, None]) ** 2).sum(1)
        delta = mean_g - mean_g.mean(0, keepdims=True)
        var2 = (m2_g.sum(0) + 32 * (delta ** 2).sum(0)) / D
        np.testing.assert_allclose(var2, var[0], rtol=1e-5)
        R = arena[off[f"lnR{i}"]:off[f"lnR{i}"] + H * 16].reshape(H, 16)[:, :D // 32]
        acc = gemm(f"pw1_{i}", (ug - mean_g[:, None]).reshape(D, T).astype(np.float32), bias=False) + R @ delta
        # round 6 default (PRE_LNP): the operand stays as it is, rstd (W' u - mean rowsum); both forms from the same arena agree
        Rs = arena[off[f"lnRs{i}"]:off[f"lnRs{i}"] + H][:, None]
        np.testing.assert_allclose(Rs[:, 0], R.sum(1), rtol=1e-5, atol=1e-6)
        acc_p = gemm(f"pw1_{i}", u.astype(np.float32), bias=False) - mu * Rs
        np.testing.assert_allclose(acc_p, acc, rtol=1e-4, atol=1e-4)
        hid = gelu((acc_p / np.sqrt(var2 + 1e-6) + arena[o1["b"]:o1["b"] + H][:, None]).astype(np.float32))
        y = gemm(f"pw2_{i}", hid)                                             # includes the bias
        np.testing.assert_allclose(gemm(f"pw2w_{i}", hid), y, rtol=1e-5, atol=1e-6)   # the 64-row-tile packing of the same weights
        X = ((X + arena[off[f"gamma{i}"]:off[f"gamma{i}"] + D][:, None] * y) * keep).astype(np.float32)
    out = gemm("out2", gelu(gemm("out0", X))) * keep
    with torch.no_grad():
        ref = convnext_ref.convnext_forward(sd, x, t, cond, mask, mask, num_layers=L, dilation_cycle=2)[0].numpy()
    np.testing.assert_allclose(out, ref, rtol=2e-4, atol=2e-5)


def test_convnext_contract_and_bad_configs(lib):
    from fish_diffusion_amd import DENOISERS, DIFFUSIONS, ConvNext
    from oracle import convnext_ref
    net = DENOISERS.build(dict(type="ConvNextDenoiser", mel_channels=16, dim=64, mlp_factor=2, condition_dim=24, num_layers=3))
    assert isinstance(net, ConvNext)
    want = [k for k, _ in convnext_ref.param_shapes(mel_channels=16, dim=64, mlp_factor=2, condition_dim=24, num_layers=3)]
    assert list(net.state_dict().keys()) == want
    assert float(net.state_dict()["residual_layers.0.gamma"][0]) == pytest.approx(1e-6)   # layer_scale_init_value, convnext.py:29
    # cross_attention=True (convnext.py:95-152,186-193): the reference's mixed residual_layers list, key for key, packable
    cx = ConvNext(mel_channels=16, dim=128, mlp_factor=2, condition_dim=24, num_layers=6, cross_attention=True)
    want_x = [k for k, _ in convnext_ref.param_shapes(mel_channels=16, dim=128, mlp_factor=2, condition_dim=24, num_layers=6, cross_every=5)]
    assert list(cx._keys) == want_x and set(cx.state_dict()) == set(want_x)
    assert "residual_layers.0.multihead_attn.in_proj_weight" in cx.state_dict() and "residual_layers.6.positional_embedding" in cx.state_dict()
    assert "residual_layers.7.gamma" in cx.state_dict() and "residual_layers.8.gamma" not in cx.state_dict()   # 2 cross + 6 ConvNeXt blocks
    from fish_diffusion_amd import _lib
    arena = _lib.pack_on_host(cx._desc, cx._params(), "convnext")
    assert np.isfinite(arena).all() and arena.size * 4 > 4096 * 128 * 4
    with pytest.raises(NotImplementedError):
        ConvNext(dim=96, cross_attention=True)              # 8 heads of 12: no attention kernel for that head size
    with pytest.raises(ValueError):
        ConvNext(dim=100)                                   # not a multiple of 32: fails at construction
    d = DIFFUSIONS.build(dict(type="GaussianDiffusion", denoiser=dict(type="ConvNextDenoiser", dim=64, num_layers=2), spec_min=[-5], spec_max=[0]))
    assert "denoise_fn.residual_layers.1.pwconv2.weight" in d.state_dict()
    with pytest.raises(RuntimeError):                       # CPU tensors: no fallback
        net(torch.zeros(1, 16, 8), torch.zeros(1), torch.zeros(1, 24, 8))


def test_tfdec_contract_and_packing(lib):
    """State-dict contract of the TransformerDecoderDenoiser mirror, bad configs, and the packed arena: the cross-attention
    in_proj is split into the query rows and the key/value rows; checked through the MFMA lane-level emulation."""
    from fish_diffusion_amd import DENOISERS, DIFFUSIONS, TransformerDecoderDenoiser
    from oracle import tfdec_ref
    cfg = dict(mel_channels=16, dim=128, mlp_factor=2, condition_dim=24, num_layers=2)
    net = DENOISERS.build(dict(type="TransformerDecoderDenoiser", **cfg))
    assert isinstance(net, TransformerDecoderDenoiser)
    assert list(net.state_dict().keys()) == [k for k, _ in tfdec_ref.param_shapes(**cfg)]
    assert torch.equal(net.positional_embedding, tfdec_ref.positional_embedding(128))
    with pytest.raises(ValueError):
        TransformerDecoderDenoiser(dim=96)                  # 8 heads of 12: not built (16 / 32 / 64 only)
    d = DIFFUSIONS.build(dict(type="GaussianDiffusion", denoiser=dict(type="TransformerDecoderDenoiser", dim=128, num_layers=1),
                              spec_min=[-5], spec_max=[0]))
    assert "denoise_fn.layers.0.multihead_attn.in_proj_weight" in d.state_dict()
    with pytest.raises(RuntimeError):                       # CPU tensors: no fallback
        net(torch.zeros(1, 16, 8), torch.zeros(1), torch.zeros(1, 24, 8))
    sd = tfdec_ref.seeded_state(3, **cfg)
    net.load_state_dict(sd, strict=True)
    arena = lib.pack_on_host(net._desc, net._params(), "tfdec")
    D, H, M, E = 128, 256, 16, 24
    r64 = lambda n: (n + 63) // 64 * 64
    cur = 128 + r64(4096 * D)
    assert arena[0] == sd["position_scale_query"][0] and arena[64] == sd["position_scale_key"][0]
    np.testing.assert_array_equal(arena[128:128 + 4096 * D].reshape(4096, D), sd["positional_embedding"].numpy())

    def plan(rows, cin, RB):
        nonlocal cur
        mt = (rows + 32 * RB - 1) // (32 * RB)
        w = mt * ((cin + 7) // 8) * RB * 64 * 4
        o = dict(w=cur, b=cur + w, mt=mt, cin8=(cin + 7) // 8, RB=RB, rows=rows)
        cur += w + r64(rows)
        return o
    for rows, cin, RB in ((H, M, 1), (D, H, 1), (H, D, 2), (D, H, 2), (H, E, 2), (D, H, 1)):   # in0 in2 emb1 emb3 cond0 cond2
        plan(rows, cin, RB)
    def plan_layer():
        nonlocal cur
        y = dict(sa_in=plan(3 * D, D, 1), sa_out=plan(D, D, 1), ca_q=plan(D, D, 1), ca_kv=plan(2 * D, D, 2), ca_out=plan(D, D, 1),
                 lin1=plan(H, D, 1), lin2=plan(D, H, 1))       # sa_in / lin1: 32-row tiles since round 6
        for k in ("n1w", "n1b", "n2w", "n2b", "n3w", "n3b"):
            y[k] = cur
            cur += r64(D)
        for k, rows in (("sa_in_R", 3 * D), ("ca_q_R", D), ("lin1_R", H)):     # row sums of the LayerNorm-folded weights (round 6)
            y[k] = cur
            cur += r64(rows)
        return y
    L0, L1 = plan_layer(), plan_layer()
    out0 = plan(D, D, 1)
    out0_R = cur
    g = torch.Generator().manual_seed(0)
    T, halo = 9, 32
    xv = torch.randn(D, T, generator=g) * 1.5 + 0.3
    Xp = np.zeros((D, halo + 64 + halo), np.float32)
    Xp[:, halo:halo + T] = xv.numpy()

    def run(o):
        acc = emulate_convgemm(arena[o["w"]:o["b"]], Xp, n_mtiles=o["mt"], RB=o["RB"], cin8=o["cin8"], taps=1, shift0=0, dshift=0, T=T)
        full = np.concatenate([acc[(mt, rb)] for mt in range(o["mt"]) for rb in range(o["RB"])])[:o["rows"]]
        return full, arena[o["b"]:o["b"] + o["rows"]][:, None]

    def run_ln(o, R_off):
        """What convgemm_kernel<.., PRE_LNP, ..> computes: rstd (W' x - mean R) + b' on the PLAIN operand."""
        acc, bias = run(o)
        mean = Xp[:, halo:halo + T].mean(0, keepdims=True)
        rstd = 1.0 / np.sqrt(Xp[:, halo:halo + T].var(0, keepdims=True) + 1e-5)
        R = arena[R_off:R_off + o["rows"]][:, None]
        return rstd * (acc - mean * R) + bias

    def ln(x, w, b):
        return torch.nn.functional.layer_norm(x.T, (D,), w, b, 1e-5).T
    # layer 0: the self-attention in-projection reads a plain tensor; key / value projection of the memory is never folded
    W, b = sd["layers.0.multihead_attn.in_proj_weight"], sd["layers.0.multihead_attn.in_proj_bias"]
    acc, bias = run(L0["ca_kv"])
    np.testing.assert_allclose(acc + bias, (W[D:] @ xv + b[D:, None]).numpy(), rtol=1e-4, atol=1e-5)
    Ws, bs = sd["layers.0.self_attn.in_proj_weight"], sd["layers.0.self_attn.in_proj_bias"]
    acc, bias = run(L0["sa_in"])
    np.testing.assert_allclose(acc + bias, (Ws @ xv + bs[:, None]).numpy(), rtol=1e-4, atol=1e-5)
    # folded consumers: query projection behind norm1, linear1 behind norm2, the NEXT layer's in-projection and output_projection.0 behind norm3
    for o, R_off, Wk, bk, nk in ((L0["ca_q"], L0["ca_q_R"], W[:D], b[:D], "layers.0.norm1"),
                                 (L0["lin1"], L0["lin1_R"], sd["layers.0.linear1.weight"], sd["layers.0.linear1.bias"], "layers.0.norm2"),
                                 (L1["sa_in"], L1["sa_in_R"], sd["layers.1.self_attn.in_proj_weight"], sd["layers.1.self_attn.in_proj_bias"], "layers.0.norm3"),
                                 (out0, out0_R, sd["output_projection.0.weight"][:, :, 0], sd["output_projection.0.bias"], "layers.1.norm3")):
        want = Wk @ ln(xv, sd[nk + ".weight"], sd[nk + ".bias"]) + bk[:, None]
        np.testing.assert_allclose(run_ln(o, R_off), want.numpy(), rtol=2e-4, atol=2e-5, err_msg=nk)
    # the norms' own parameters stay in the arena: the residual epilogue (EpiResLN) applies them to the old value
    np.testing.assert_array_equal(arena[L0["n2w"]:L0["n2w"] + D], sd["layers.0.norm2.weight"].numpy())
    np.testing.assert_array_equal(arena[L1["n3b"]:L1["n3b"] + D], sd["layers.1.norm3.bias"].numpy())


# ------------------------------------------------------------------ opt-in bf16 storage mode: packing + blocked operand layout
def test_wavenet_bf16_packing_and_blocked_layout_match_kernel_index_math(lib):
    """numpy emulation of the bf16 K loop of convgemm_kernel<..., OPK_BF16> (v_mfma_f32_32x32x16_bf16 operand maps: lane (half, i)
    holds row / column i and the 8 consecutive k = 8*half .. 8*half+7) over the arena written by fdx_wavenet_bf16_pack and a
    C8-blocked bf16 activation buffer, against F.conv1d on the same bf16-rounded operands."""
    from fish_diffusion_amd import WaveNet
    net = WaveNet(**WN_SMALL)
    sd = wavenet_sd(WN_SMALL, 3)
    net.load_state_dict(sd)
    nb = C.c_size_t()
    lib.check(lib.lib().fdx_wavenet_bf16_packed_bytes(C.byref(net._desc), C.byref(nb)))
    keep, arr = lib.host_ptr_array(net._params())
    raw = np.zeros(nb.value // 2, np.uint16)
    lib.check(lib.lib().fdx_wavenet_bf16_pack(C.byref(net._desc), arr, len(keep), C.c_void_p(raw.ctypes.data), nb))
    arena = (raw.astype(np.uint32) << 16).view(np.float32)                     # bf16 -> fp32, exact
    Cc, L = 64, 4
    conv_mt, outp_mt, conv_it, outp_it = Cc // 32, 2 * Cc // 64, (Cc // 16) * 3, Cc // 16
    per_layer = (conv_mt * conv_it + outp_mt * outp_it) * 2 * 64 * 8
    assert arena.size == L * per_layer
    rb16 = lambda t: t.to(torch.bfloat16).to(torch.float32)
    T, halo, ld = 40, 32, 32 + 64 + 32
    g = torch.Generator().manual_seed(0)
    y = rb16(torch.randn(Cc, T, generator=g))
    Yb = np.zeros((Cc // 8, ld, 8), np.float32)                                # element (c, t) at [c >> 3][halo + t][c & 7]
    for c in range(Cc):
        Yb[c >> 3, halo:halo + T, c & 7] = y[c].numpy()

    def contract(frags, n_it, taps, shift0, dshift):                           # frags [n_it][2 rb][64 lanes][8]; one 64-column tile at t0 = 0
        acc = np.zeros((2, 2, 32, 32))                                         # [rb][nb][row i][column n]
        n = np.arange(32)
        for it in range(n_it):
            cb16, tap = divmod(it, taps)
            for half in range(2):
                for nbk in range(2):
                    col = halo + 2 * n + nbk + shift0 + tap * dshift            # lane n owns the column pair (2n, 2n+1)
                    Bk = Yb[2 * cb16 + half][col]                               # [32 columns][8 k]
                    for rb in range(2):
                        A = frags[it, rb, half * 32:half * 32 + 32]             # [32 rows][8 k]
                        acc[rb, nbk] += A.astype(np.float64) @ Bk.T.astype(np.float64)
        out = np.zeros((2, 32, 64))
        out[:, :, 0::2], out[:, :, 1::2] = acc[:, 0], acc[:, 1]
        return out[:, :, :T]

    layer, dil = 1, 2                                                           # layer 1: dilation 2
    base = layer * per_layer
    conv = arena[base:base + conv_mt * conv_it * 2 * 64 * 8].reshape(conv_mt, conv_it, 2, 64, 8)
    W = rb16(sd[f"residual_layers.{layer}.conv_layer.conv.weight"])
    ref = F.conv1d(y[None].double(), W.double(), None, padding=dil, dilation=dil)[0].numpy()
    for mt in range(conv_mt):
        got = contract(conv[mt], conv_it, 3, -dil, dil)
        np.testing.assert_allclose(got[0], ref[mt * 32:mt * 32 + 32], rtol=1e-6, atol=1e-6)                    # gate rows
        np.testing.assert_allclose(got[1], ref[Cc + mt * 32:Cc + mt * 32 + 32], rtol=1e-6, atol=1e-6)          # filter rows
    o0 = base + conv_mt * conv_it * 2 * 64 * 8
    outp = arena[o0:o0 + outp_mt * outp_it * 2 * 64 * 8].reshape(outp_mt, outp_it, 2, 64, 8)
    Wo = rb16(sd[f"residual_layers.{layer}.output_projection.conv.weight"][:, :, 0])
    refo = (Wo.double() @ y.double()).numpy()
    for mt in range(outp_mt):
        got = contract(outp[mt], outp_it, 1, 0, 0)
        np.testing.assert_allclose(np.concatenate([got[0], got[1]]), refo[mt * 64:mt * 64 + 64], rtol=1e-6, atol=1e-6)
    # the storage switch itself is host state until a device exists
    net.storage = "bf16"
    assert net.storage == "bf16"
    with pytest.raises(ValueError):
        net.storage = "fp16"


def test_xcd_rect_tile_map_is_a_bijection():
    """csrc/convgemm.hip.h `conv_tile_of_block` / `conv_rect_grid` (mirrored here, and the mirror is read against the source): block b runs
    on XCD b % 8; with a rectangle map XCD x owns one of 2 row halves x 4 column quarters (mode 1) or 4 row quarters x 2 column halves
    (mode 2), walked row-fastest.  Every logical tile exactly once, every XCD inside its rectangle, and the tiles an XCD has in flight at
    once (32 consecutive slots) cover all of its row tiles.  Round 6: the column tiles need not divide by the column groups -- the launch
    is padded to 8 equal rectangles and the workgroups past the end return (the serving micro-batches' 93 x 16 grids)."""
    src = open(f"{ROOT}/fish_diffusion_amd/csrc/convgemm.hip.h").read()
    for line in ("const int MH = n_mt >> rs, QC = 8 >> rs, NQ = (n_tiles_n + QC - 1) / QC;", "nt = (xcd >> rs) * NQ + ntl;", "return nt < n_tiles_n;",
                 "return 8 * (n_mt >> rs) * ((n_tiles_n + QC - 1) / QC);"):
        assert line in src, line

    def rect_grid(n_tiles_n, n_mt, rect):
        if not rect:
            return n_tiles_n * n_mt
        rs = 2 if rect == 2 else 1
        QC = 8 >> rs
        return 8 * (n_mt >> rs) * ((n_tiles_n + QC - 1) // QC)

    def tile_of_block(n_tiles_n, n_mt, rect, bid):
        G, xcd, slot = n_tiles_n * n_mt, bid & 7, bid >> 3
        if rect:
            rs = 2 if rect == 2 else 1
            MH, QC = n_mt >> rs, 8 >> rs
            NQ = (n_tiles_n + QC - 1) // QC
            ntl = slot // MH
            m, n = (xcd & ((1 << rs) - 1)) * MH + (slot - ntl * MH), (xcd >> rs) * NQ + ntl
            return (m, n) if n < n_tiles_n else None
        q8, r8 = G >> 3, G & 7
        L = (xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8) + slot
        return L // n_tiles_n, L % n_tiles_n
    for rect in (1, 2):
        rs = rect
        QC = 8 >> rs
        for n_tiles_n, n_mt in ((224, 16), (128, 32), (4, 4), (12, 16), (93, 16), (104, 16), (7, 4), (5, 8), (2, 4), (33, 12)):
            if n_tiles_n < QC:
                continue
            G = rect_grid(n_tiles_n, n_mt, rect)
            NQ = (n_tiles_n + QC - 1) // QC
            tiles = [tile_of_block(n_tiles_n, n_mt, rect, b) for b in range(G)]
            live = [t for t in tiles if t is not None]
            assert len(live) == len(set(live)) == n_tiles_n * n_mt and set(live) == {(m, n) for m in range(n_mt) for n in range(n_tiles_n)}
            assert G - len(live) == (QC * NQ - n_tiles_n) * n_mt        # the padding: whole column tiles of the last group(s)
            for b, t in enumerate(tiles):
                if t is None:
                    continue
                m, n = t
                x = b & 7
                assert m // (n_mt >> rs) == (x & ((1 << rs) - 1)) and n // NQ == (x >> rs)
            if G // 8 >= 32 and (n_mt >> rs) <= 32:
                first = {tile_of_block(n_tiles_n, n_mt, rect, 8 * s)[0] for s in range(32)}
                assert first == set(range(n_mt >> rs))
    for n_tiles_n, n_mt in ((14, 16), (8, 32), (7, 3)):      # the row-run map, any shape
        G = n_tiles_n * n_mt
        assert {tile_of_block(n_tiles_n, n_mt, 0, b) for b in range(G)} == {(m, n) for m in range(n_mt) for n in range(n_tiles_n)}


def test_bcast_arena_entry_validates_without_a_gpu(lib):
    """`fdx_bcast_arena` (SURVEY 8(b)'s fdx_bcast_weights): exported, bound to RCCL only at call time, and a null arena / communicator is an
    argument error -- no GPU, no RCCL needed to find that out (the collective itself: tests/test_gpu_round6.py, a world-1 communicator)."""
    import ctypes as C
    from fish_diffusion_amd import _lib
    with pytest.raises(ValueError, match="fdx_bcast_arena"):
        _lib.check(_lib.lib().fdx_bcast_arena(None, 16, None, 0, None))
    buf = (C.c_char * 16)()
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().fdx_bcast_arena(C.cast(buf, C.c_void_p), 16, None, 0, None))
