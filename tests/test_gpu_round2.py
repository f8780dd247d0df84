"""GPU tests added in round 2: full-size hop-256 vocoder golden, chained features -> waveform parity against the reference chain
and its fp64 yardstick, the shallow-diffusion entry in the library, the DDPM sampler's chunked reference-RNG noise and clip
bounds, the bf16 arena derived on the device, the recorded-graph cache under ragged serving, and the weight broadcast onto a rank
whose own parameters differ."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.helpers import ROOT, WN_FULL, WN_SMALL, abs_err, load, rel_err, sha1_state, synth_f0, wavenet_sd

pytestmark = pytest.mark.gpu

MEL_REL = 1e-3   # north_star: 1e-3 rel fp32 on mel
WAV_ABS = 1e-4   # north_star: 1e-4 abs on waveform samples


@pytest.fixture(scope="module")
def dev(lib_built):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _diffusion(cfg, sd, dev, **kw):
    from fish_diffusion_amd import DIFFUSIONS
    d = DIFFUSIONS.build(dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", **cfg), spec_min=[-5], spec_max=[0], **kw))
    d.denoise_fn.load_state_dict(sd, strict=True)
    return d.to(dev).eval()


def _vocoder(h, gsd, dev, **kw):
    from fish_diffusion_amd import NsfHifiGAN
    return NsfHifiGAN.from_state(h, gsd, **kw).to(dev)


def _regen_source_noise(g, L):
    torch.manual_seed(int(g["noise_seed"]))          # regenerate the injected draws (too big to store), verify by SHA-1
    rand_ini = torch.rand(1, 9)
    rand_ini[:, 0] = 0
    src_noise = torch.randn(1, L, 9)
    assert hashlib.sha1(src_noise.numpy().tobytes()).hexdigest() == str(g["src_noise_sha1"])
    assert torch.equal(rand_ini, g["rand_ini"])
    return rand_ini, src_noise


# ------------------------------------------------------------------------------------------------ configs[2]: hop 256, full size
def test_generator_hop256_full_10s_matches_reference_golden_and_batch32(dev):
    """tools/nsf_hifigan/config_v1_256.json (what configs/vocoder_nsf_hifigan.py:9,31 points at) at BASELINE configs[2]'s size: 10 s,
    T = 1722 -> 440 832 samples, vs the real `Generator`'s output; then the same item as every member of a batch of 32 (configs[2]'s
    batch) must reproduce the batch-1 waveform bit for bit (tile counts and the split-K / no-split kernel choice differ with B)."""
    from oracle import nsf_hifigan_ref
    g = load("nsf_v1_256_full")
    h = json.loads(str(g["config"]))
    assert h["hop_size"] == 256 and g["mel"].shape[-1] == 1722
    gsd = nsf_hifigan_ref.seeded_generator_state(int(g["seed"]), h)
    assert sha1_state(gsd) == str(g["weights_sha1"])
    T = g["mel"].shape[-1]
    rand_ini, src_noise = _regen_source_noise(g, T * 256)
    voc = _vocoder(h, gsd, dev)
    mel, f0 = g["mel"].to(dev), g["f0"].to(dev)
    ri, sn = rand_ini.to(dev), src_noise.to(dev)
    wav = voc.model(mel, f0, rand_ini=ri, src_noise=sn)
    err = abs_err(wav.cpu(), g["wav"])
    print(f"hop 256, full 10 s: wav abs err {err:.3e}  (peak |wav| {float(g['wav'].abs().max()):.3f})")
    assert wav.shape == (1, 1, 440832) and err < WAV_ABS
    B = 32
    wb = voc.model(mel.expand(B, -1, -1).contiguous(), f0.expand(B, -1).contiguous(), rand_ini=ri.expand(B, -1).contiguous(),
                   src_noise=sn.expand(B, -1, -1).contiguous())
    assert wb.shape == (B, 1, 440832)
    eb = abs_err(wb[0].cpu(), g["wav"][0])
    spread = float((wb - wb[:1]).abs().max())
    print(f"batch 32: item 0 vs golden {eb:.3e}, spread across identical items {spread:.3e}, vs batch-1 run {abs_err(wb[0].cpu(), wav[0].cpu()):.3e}")
    assert eb < WAV_ABS and spread == 0.0


# ------------------------------------------------------------------------------------------------ chained parity
@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_chained_features_to_waveform_vs_reference_chain_and_fp64(dev, tag):
    """The chain tools/diffusion/inference.py:140-160 runs (features -> UniPC sampler -> mel -> vocoder -> waveform) end to end on the
    device against (a) the real reference's fp32 chain and (b) the same chain with an fp64 data path (oracle/make_golden.py
    `golden_round2`).  Per stage the north_star bars hold with two orders of margin (the tests above).  Chained, the vocoder
    amplifies 1e-6-class mel noise, and the REFERENCE's own fp32 chain sits 6.5e-5 / 8.7e-5 from the exact result -- so the
    claim that can be made of any fp32 implementation, and is asserted here, is: HIP is as close to the exact (fp64) waveform as
    the reference is, and within the sum of the two fp32 errors of the reference."""
    from oracle import nsf_hifigan_ref
    g = load(f"chain_{tag}")
    sd = wavenet_sd(WN_FULL, 1234)
    assert sha1_state(sd) == str(g["wn_sha1"])
    diff = _diffusion(WN_FULL, sd, dev)
    h = nsf_hifigan_ref.CONFIG_V1
    gsd = nsf_hifigan_ref.seeded_generator_state(55, h)
    assert sha1_state(gsd) == str(g["voc_sha1"])
    voc = _vocoder(h, gsd, dev, use_natural_log=False)
    T = g["features"].shape[1]
    rand_ini, src_noise = _regen_source_noise(g, T * 512)
    mel = diff(g["features"].to(dev), sampler_interval=int(g["interval"]), x_init=g["x_init"].to(dev))
    wav = voc.model(mel.transpose(1, 2).contiguous(), g["f0"].to(dev), rand_ini=rand_ini.to(dev), src_noise=src_noise.to(dev),
                    mel_scale=2.30259).cpu()
    mel = mel.cpu()
    e_mel_ref, e_mel_64 = rel_err(mel, g["mel"]), rel_err(mel, g["mel64"])
    e_ref, e_64 = abs_err(wav, g["wav"]), abs_err(wav, g["wav64"])
    r_64, r_mel = float(g["ref_vs_f64_wav_abs"]), float(g["ref_vs_f64_mel_rel"])
    print(f"chain {tag}: mel rel  HIP-ref {e_mel_ref:.2e}  HIP-f64 {e_mel_64:.2e}  ref-f64 {r_mel:.2e} | "
          f"wav abs  HIP-ref {e_ref:.2e}  HIP-f64 {e_64:.2e}  ref-f64 {r_64:.2e}")
    assert e_mel_ref < MEL_REL and e_mel_64 < MEL_REL
    assert e_64 <= 1.25 * r_64, "HIP is further from the exact waveform than the reference's own fp32 chain"
    assert e_ref <= e_64 + r_64 + 1e-7 and e_ref < 2 * WAV_ABS
    if e_ref >= WAV_ABS:   # only reachable when the reference itself is that far from exact
        assert r_64 > 0.5 * WAV_ABS


# ------------------------------------------------------------------------------------------------ a7 in the library
def test_shallow_diffusion_entry_runs_in_the_library(dev):
    """diffusion.py:223-232 through fdx_q_sample: norm_spec + q_sample with the reference's draw order == the oracle expression;
    per-frame stats broadcast over the LAST axis like the reference's expression; the forward with original_mel / skip_steps equals
    a run started from the explicitly built x_T."""
    from fish_diffusion_amd import _lib
    from oracle import sampler_ref
    diff = _diffusion(WN_SMALL, wavenet_sd(WN_SMALL, 101), dev)
    g = torch.Generator().manual_seed(4)
    B, M, T = 2, 128, 20
    mel0 = -5 * torch.rand(B, M, T, generator=g)
    nz = torch.randn(B, M, T, generator=g)
    eng = diff.denoise_fn.engine(dev)
    out = diff._shallow_init(eng, mel0.to(dev), True, 400, nz.to(dev), _lib.stream_ptr(dev)).cpu()
    smin, smax = torch.tensor([-5.0]).view(1, 1, 1), torch.tensor([0.0]).view(1, 1, 1)
    ref = sampler_ref.q_sample(sampler_ref.norm_spec(mel0, smin, smax), 600, nz, sampler_ref.beta_schedule())
    assert torch.equal(out, ref), abs_err(out, ref)
    only_norm = diff._shallow_init(eng, mel0.to(dev), True, 0, None, _lib.stream_ptr(dev)).cpu()
    assert torch.equal(only_norm, sampler_ref.norm_spec(mel0, smin, smax))
    # end to end: the module draws randn_like(x) after seeding, exactly like the reference (diffusion.py:232)
    feats = torch.randn(B, T, 256, generator=g).to(dev)
    torch.manual_seed(9)
    a = diff(feats, sampler_interval=100, skip_steps=400, original_mel=mel0.to(dev))
    torch.manual_seed(9)
    noise = torch.randn_like(mel0.to(dev))
    x_T = sampler_ref.q_sample(sampler_ref.norm_spec(mel0, smin, smax), 600, noise.cpu(), sampler_ref.beta_schedule())
    b = diff(feats, sampler_interval=100, skip_steps=400, x_init=x_T.to(dev))
    assert torch.equal(a, b)
    # [1,1,n] statistics broadcast against the last axis of [B,M,T] in the reference's expression: n must be 1 or T
    from fish_diffusion_amd import DIFFUSIONS
    d2 = DIFFUSIONS.build(dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-6.0] * 128,
                               spec_max=[0.5] * 128)).to(dev).eval()
    with pytest.raises(RuntimeError):
        d2(feats, sampler_interval=100, skip_steps=400, original_mel=mel0.to(dev))


# ------------------------------------------------------------------------------------------------ DDPM noise stream, clip bounds
def test_naive_sampler_chunked_reference_rng_and_clip_buffers(dev):
    """(a) step_rng = "torch": one randn_like per step in the reference's order (noise_predictor.py:101), drawn a bounded chunk of
    steps ahead -- bit-identical to handing over the same draws as one [n_steps, B, M, T] tensor, whatever the chunk size;
    (b) naive_noise_predictor.clip_min / clip_max are honoured (noise_predictor.py:30-31,90: loadable buffers)."""
    diff = _diffusion(WN_SMALL, wavenet_sd(WN_SMALL, 101), dev)
    g = torch.Generator().manual_seed(6)
    B, T = 2, 33
    feats, x0 = torch.randn(B, T, 256, generator=g).to(dev), torch.randn(B, 128, T, generator=g).to(dev)
    kw = dict(sampler_interval=20, noise_predictor="naive", x_init=x0)     # 50 steps
    torch.manual_seed(123)
    noise = torch.stack([torch.randn_like(x0) for _ in range(50)])
    ref = diff(feats, step_noise=noise, **kw)
    for chunk_bytes in (1, 7 * B * 128 * T * 4, 1 << 30):                   # 1 step per chunk, 7 steps per chunk, everything at once
        diff.naive_noise_chunk_bytes = chunk_bytes
        torch.manual_seed(123)
        out = diff(feats, **kw)
        assert torch.equal(out, ref), chunk_bytes
    diff.naive_noise_chunk_bytes = 128 << 20
    # clip bounds: a tighter clamp must change the result, and equal the oracle with the same bounds
    from oracle import sampler_ref
    from tests.test_gpu_parity import _oracle_den
    diff.naive_noise_predictor.clip_min.fill_(-0.5)
    diff.naive_noise_predictor.clip_max.fill_(0.25)
    out = diff(feats, step_noise=noise, **kw)
    assert not torch.equal(out, ref)
    den = _oracle_den(wavenet_sd(WN_SMALL, 101), WN_SMALL)
    tb = sampler_ref.NaiveTables(sampler_ref.beta_schedule())
    x = x0.cpu()
    with torch.no_grad():
        for i, t in enumerate(list(range(0, 1000, 20))[::-1]):
            eps = den(x, torch.full((1,), t, dtype=torch.long), feats.cpu().transpose(1, 2), None, None)
            x0p = torch.clamp(tb.sqrt_recip[t] * x - tb.sqrt_recipm1[t] * eps, min=-0.5, max=0.25)
            x = tb.coef1[t] * x0p + tb.coef2[t] * x + (1.0 if t > 0 else 0.0) * (0.5 * tb.logvar[t]).exp() * noise[i].cpu()
    want = sampler_ref.denorm_spec(x.transpose(1, 2), torch.tensor([-5.0]).view(1, 1, 1), torch.tensor([0.0]).view(1, 1, 1))
    assert rel_err(out.cpu(), want) < MEL_REL


# ------------------------------------------------------------------------------------------------ bf16 arena from the fp32 arena
def test_bf16_arena_derived_on_device_equals_host_pack(dev):
    """fdx_wavenet_bf16_from_arena (what storage = "bf16" uses: the ATTACHED fp32 arena, not the module's parameters) must produce
    byte for byte what fdx_wavenet_bf16_pack produces on the host from the original tensors (the arena holds the dilated conv in the
    16x16x4 fragment order and the out-projection in the 32x32x2 one: both derivations are exercised)."""
    code = r'''
import ctypes as C, sys, torch
sys.path.insert(0, %r)
from fish_diffusion_amd import DENOISERS, _lib
from tests.helpers import WN_SMALL, wavenet_sd
dev = torch.device("cuda", 0)
cfg = dict(WN_SMALL, residual_channels=128, residual_layers=3)
net = DENOISERS.build(dict(type="WaveNetDenoiser", **cfg))
torch.manual_seed(5)
for p in net.parameters():
    torch.nn.init.normal_(p, std=0.3)
net = net.to(dev).eval()
net.storage = "bf16"
eng = net.engine(dev)
torch.cuda.synchronize()
l = _lib.lib()
nb = C.c_size_t()
_lib.check(l.fdx_wavenet_bf16_packed_bytes(C.byref(net._desc), C.byref(nb)))
keep, arr = _lib.host_ptr_array(net._params())
host = torch.empty(nb.value, dtype=torch.uint8)
_lib.check(l.fdx_wavenet_bf16_pack(C.byref(net._desc), arr, len(keep), C.c_void_p(host.data_ptr()), nb))
got = net._arena_bf16.cpu()
assert got.numel() == host.numel() and torch.equal(got, host), int((got != host).sum())
print("OK", nb.value)
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


# ------------------------------------------------------------------------------------------------ ragged serving vs the graph cache
def test_ragged_stream_of_40_lengths_recaptures_nothing_after_warmup(dev):
    """tools/diffusion/inference.py:336-376 feeds segments of arbitrary length.  pipeline.synthesize pads every micro-batch to a
    64-frame bucket and the library keeps an LRU of recorded sampler graphs, so a stream of >= 32 distinct lengths settles on a
    handful of geometries: after one pass over the stream, a second pass captures NOTHING.  With `exact=False` (the reference's
    padded-batch semantics) bucketed results equal the oracle run of the same padded, masked batch; the default exact-ragged mode
    is covered by test_exact_ragged_batches_equal_batch_one_runs_bit_for_bit."""
    from fish_diffusion_amd import pipeline
    from oracle import nsf_hifigan_ref, sampler_ref
    from tests.test_gpu_parity import _oracle_den
    sd = wavenet_sd(WN_SMALL, 101)
    diff = _diffusion(WN_SMALL, sd, dev)
    h = dict(nsf_hifigan_ref.CONFIG_V1)
    voc = _vocoder(h, nsf_hifigan_ref.seeded_generator_state(78, h), dev, use_natural_log=False)
    voc.model.rng = "philox"
    g = torch.Generator().manual_seed(40)
    lens = sorted(set(torch.randint(40, 300, (64,), generator=g).tolist()))[:40]
    assert len(lens) == 40
    perm = torch.randperm(40, generator=g).tolist()
    lens = [lens[i] for i in perm]
    feats = [torch.randn(n, 256, generator=g).to(dev) for n in lens]
    f0s = [synth_f0(n).to(dev) for n in lens]
    eng = diff.denoise_fn.engine(dev)

    def one_pass():
        out = []
        for k in range(0, 40, 4):      # a serving loop: requests arrive four at a time
            out += pipeline.synthesize(diff, voc, feats[k:k + 4], f0s[k:k + 4], max_batch=2, sampler_interval=250)
        return out
    one_pass()
    c1, l1, n1 = eng.graph_stats()
    res = one_pass()
    c2, l2, n2 = eng.graph_stats()
    print(f"graphs: captured {c1} in the first pass, {c2 - c1} in the second; launches {l1} -> {l2}; cached {n2}")
    assert c2 == c1 and l2 == 2 * l1 and n2 <= 48 and c1 <= 10      # batch sizes {1, 2} x buckets {64 .. 320}
    assert len(res) == 40 and all(torch.isfinite(w).all() for _, _, w in res)
    # bucketed + masked == the oracle on the same padded batches (same sharding / batching decisions, reference semantics)
    from fish_diffusion_amd.dist import shard_utterances
    sel = [i for i in range(40) if 64 < lens[i] <= 192][:5]
    sl = [lens[i] for i in sel]
    x_all = torch.randn(len(sel), 128, 192, generator=g)
    r = pipeline.synthesize(diff, voc, [feats[i] for i in sel], [f0s[i] for i in sel], max_batch=2, sampler_interval=100, exact=False,
                            x_init_fn=lambda ii, M, T: torch.stack([x_all[i, :, :T] for i in ii]).to(dev))
    by = {i: m.cpu() for i, m, _ in r}
    mine = shard_utterances(sl, 0, 1)
    den = _oracle_den(sd, WN_SMALL)
    for group in pipeline.make_batches([sl[i] for i in mine], 2):
        ii = [mine[k] for k in group]
        T = (max(sl[i] for i in ii) + 63) // 64 * 64
        fb = torch.zeros(len(ii), T, 256)
        for b, i in enumerate(ii):
            fb[b, :sl[i]] = feats[sel[i]].cpu()
        masks = torch.arange(T)[None] >= torch.tensor([sl[i] for i in ii])[:, None]
        assert masks.any()                        # no length is a multiple of 64 here: even the longest member is padded
        with torch.no_grad():
            ref = sampler_ref.diffusion_sample(den, fb, x_init=torch.stack([x_all[i, :, :T] for i in ii]), sampler_interval=100,
                                               x_masks=masks, cond_masks=masks)
        for b, i in enumerate(ii):
            assert rel_err(by[i], ref[b, :sl[i]]) < MEL_REL, (i, sl[i], T)


# ------------------------------------------------------------------------------------------------ broadcast onto a rank with other weights
def test_two_ranks_broadcast_arena_onto_rank_with_different_parameters(dev):
    """dist.broadcast_model_weights: a non-source rank's own parameters are never used -- not by the fp32 kernels and not by the
    bf16 storage mode (round-1 advisor finding: it re-packed from the module's own parameters).  Two processes share this GPU;
    the collective runs over gloo (RCCL refuses two ranks on one device -- the RCCL leg is covered by
    `FDX_FORCE_PROCESS_GROUP=1 python bench.py`, profiles/), the attach path is the same."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from fish_diffusion_amd import DIFFUSIONS, dist as fdist
from tests.helpers import WN_SMALL
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
dev = torch.device("cuda", 0)
torch.manual_seed(100 + rank)                     # DIFFERENT parameters on the two ranks
diff = DIFFUSIONS.build(dict(type="GaussianDiffusion", denoiser=dict(type="WaveNetDenoiser", **WN_SMALL), spec_min=[-5], spec_max=[0]))
for p in diff.denoise_fn.parameters():
    torch.nn.init.normal_(p, std=0.05)
diff = diff.to(dev).eval()
fdist.broadcast_model_weights(diff.denoise_fn, None, dev, src=0)
g = torch.Generator().manual_seed(3)
feats, x0 = torch.randn(2, 40, 256, generator=g).to(dev), torch.randn(2, 128, 40, generator=g).to(dev)
outs = []
for storage in ("fp32", "bf16", "fp32"):
    diff.denoise_fn.storage = storage
    outs.append(diff(feats, sampler_interval=100, x_init=x0).cpu())
assert torch.equal(outs[0], outs[2])
both = [torch.empty_like(torch.stack(outs)) for _ in range(2)]
dist.all_gather(both, torch.stack(outs))
assert torch.equal(both[0], both[1]), "rank 1 did not compute with rank 0's weights"
assert not torch.equal(both[0][0], both[0][1])     # bf16 mode really is a different computation
if rank == 0:
    print("OK")
dist.destroy_process_group()
''' % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[0] + o[1] for o in outs)
    assert "OK" in outs[0][0]


# ------------------------------------------------------------------------------------------------ shape-adaptive tiles
def test_every_conv_tile_shape_is_bit_identical_to_the_64x64_tile(dev):
    """convgemm16s.hip.h: the workgroup tile of the dilated conv + gate (NR 16-row blocks x NM*16 columns) is a scheduling choice --
    every output element is the same k-ordered fp32 fma chain over the same four K ranges.  Each of the nine non-default shapes,
    forced through FDX_CONV_SHAPE in its own process, must reproduce the fixed 64 x 64 tile's ("44": the round-1 kernel) result BIT FOR BIT on ragged
    geometries (tile overhang, T < one tile, odd T, batch > 1, masks), and the automatic choice must as well."""
    code = r'''
import os, sys, hashlib, torch
sys.path.insert(0, %r)
from fish_diffusion_amd import DENOISERS
from tests.helpers import WN_SMALL, wavenet_sd
dev = torch.device("cuda", 0)
cfg = dict(WN_SMALL, residual_channels=128, residual_layers=5)
net = DENOISERS.build(dict(type="WaveNetDenoiser", **cfg))
torch.manual_seed(5)
for p in net.parameters():
    torch.nn.init.normal_(p, std=0.08)
net = net.to(dev).eval()
h = hashlib.sha1()
g = torch.Generator().manual_seed(1)
for B, T in ((1, 1), (1, 37), (2, 113), (1, 257), (3, 430), (1, 861)):
    x, c, t = torch.randn(B, 128, T, generator=g).to(dev), torch.randn(B, 256, T, generator=g).to(dev), (torch.rand(B, generator=g) * 999).to(dev)
    m = torch.zeros(B, T, dtype=torch.bool, device=dev)
    m[-1, T - T // 5:] = True
    for masks in (None, m):
        y = net(x, t, c, x_masks=masks, cond_masks=masks)
        assert torch.isfinite(y).all()
        h.update(y.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
''' % ROOT
    digests = {}
    # (round 3) the 2-D tile -> XCD map of large grids (FDX_XCD_RECT=<min grid>) forced from 8 workgroups up is one more scheduling choice
    env = dict(os.environ, FDX_XCD_RECT="8")
    env.pop("FDX_CONV_SHAPE", None)
    env.pop("FDX_OUTP_SHAPE", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIGEST" in r.stdout, "xcd_rect\n" + r.stdout + r.stderr
    digests["xcd_rect"] = r.stdout.split("DIGEST")[1].split()[0]
    # (round 3) the out-projection's shape is forced alongside (FDX_OUTP_SHAPE), incl. the 16-row tiles (NR = 1) that only it has
    outp = {"44": "44", "auto": None, "45": "45", "46": "14", "47": "15", "48": "16", "24": "17", "25": "18", "26": "24", "27": "27", "28": "28"}
    for shape in ("44", "auto", "45", "46", "47", "48", "24", "25", "26", "27", "28"):
        env = dict(os.environ)
        env.pop("FDX_CONV_SHAPE", None)
        env.pop("FDX_OUTP_SHAPE", None)
        if shape != "auto":
            env["FDX_CONV_SHAPE"] = shape
            env["FDX_OUTP_SHAPE"] = outp[shape]
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "DIGEST" in r.stdout, shape + "\n" + r.stdout + r.stderr
        digests[shape] = r.stdout.split("DIGEST")[1].split()[0]
    print(digests)
    assert len(set(digests.values())) == 1, digests


# ------------------------------------------------------------------------------------------------ ConvNext(cross_attention=True)
def _cnx(cfg, sd, dev):
    from fish_diffusion_amd import DENOISERS
    net = DENOISERS.build(dict(type="ConvNextDenoiser", cross_attention=True, cross_every_n_layers=5, **cfg))
    net.load_state_dict(sd, strict=True)
    return net.to(dev).eval()


@pytest.mark.parametrize("tag", ["small", "full"])
def test_convnext_cross_attention_forward_matches_reference_golden(dev, tag):
    """SURVEY 8f row 4, the branch round 1 left out: fish_diffusion/modules/convnext.py:95-152 (CrossAttentionBlock in front of every
    5th ConvNeXt block, :186-193; the ConvNeXt blocks then run without the condition term, :246-250) against the REAL module's
    outputs: plain, masked (key-padding masks on both attentions), long t, 4-D input."""
    from tests.test_oracle_golden import CNX_FULL, CNX_SMALL, _cnx_den, _cnx_sd
    cfg = CNX_SMALL if tag == "small" else CNX_FULL
    g = load(f"convnext_cross_{tag}")
    sd = _cnx_sd(cfg, int(g["seed"]))
    assert sha1_state({k: v for k, v in sd.items() if not k.endswith("positional_embedding")}) == str(g["weights_sha1"])
    net = _cnx(cfg, sd, dev)
    x, cond, t, m = g["x"].to(dev), g["cond"].to(dev), g["t"].to(dev), g["masks"].bool().to(dev)
    eps = net(x, t, cond)
    print(f"convnext cross {tag}: eps rel err {rel_err(eps.cpu(), g['eps']):.3e}")
    assert rel_err(eps.cpu(), g["eps"]) < 2e-5
    eps_m = net(x, t, cond, x_masks=m, cond_masks=m)
    assert rel_err(eps_m.cpu(), g["eps_masked"]) < 2e-5
    assert (eps_m[1, :, g["masks"][1].bool()] == 0).all()
    assert rel_err(net(x, torch.tensor([400], device=dev), cond).cpu(), g["eps_long"]) < 2e-5
    eps4 = net(x[:, None], t, cond)
    assert eps4.shape == (x.shape[0], 1, 128, x.shape[2]) and torch.equal(eps4[:, 0], eps)
    if tag == "small":   # ragged lengths vs the oracle, masks on one side only
        den = _cnx_den(sd, cfg)
        for B, T in ((1, 1), (3, 7), (2, 65), (1, 257)):
            gg = torch.Generator().manual_seed(T)
            xx, cc, tt = torch.randn(B, 128, T, generator=gg), torch.randn(B, 256, T, generator=gg), torch.rand(B, generator=gg) * 999
            xm = torch.zeros(B, T, dtype=torch.bool)
            xm[-1, T - T // 3:] = True
            with torch.no_grad():
                ref, ref_m = den(xx, tt, cc, None, None), den(xx, tt, cc, None, xm)
            assert rel_err(net(xx.to(dev), tt.to(dev), cc.to(dev)).cpu(), ref) < 2e-5, (B, T)
            if T > 1:
                assert rel_err(net(xx.to(dev), tt.to(dev), cc.to(dev), cond_masks=xm.to(dev)).cpu(), ref_m) < 2e-5, (B, T)


@pytest.mark.parametrize("name", ["unipc_i50", "plms_i50"])
def test_sampler_over_convnext_cross_attention_matches_reference_golden(dev, name):
    """GaussianDiffusion driving ConvNext(cross_attention=True): the hoisted per-block keys / values, PLMS's one unmasked call (its
    attention memory is the UNMASKED condition, diffusion.py:285), recorded graphs (second run)."""
    from fish_diffusion_amd import DIFFUSIONS
    from tests.test_oracle_golden import CNX_SMALL, _cnx_sd
    g = load(f"convnext_cross_sampler_small_{name}")
    diff = DIFFUSIONS.build(dict(type="GaussianDiffusion", denoiser=dict(type="ConvNextDenoiser", cross_attention=True, **CNX_SMALL),
                                 spec_min=[-5], spec_max=[0]))
    diff.denoise_fn.load_state_dict(_cnx_sd(CNX_SMALL, 311), strict=True)
    diff = diff.to(dev).eval()
    m = g["masks"].bool().to(dev)
    for _ in range(2):
        mel = diff(g["features"].to(dev), sampler_interval=int(g["interval"]), noise_predictor=name.split("_")[0], x_masks=m, cond_masks=m,
                   x_init=g["x_init"].to(dev))
        assert rel_err(mel.cpu(), g["mel"]) < MEL_REL, name


# ------------------------------------------------------------------------------------------------ RefineGAN, sine template
@pytest.mark.parametrize("tag", ["small", "long"])
def test_refinegan_sine_template_matches_reference_golden(dev, tag):
    """RefineGANGenerator(template_generator="sine") (generator.py:324,338-339; SineGen :197-310) vs the real module: the template is
    the NSF source module's kernels with one harmonic + the Nyquist clean-up (the `long` fixture holds f0 > sr // 2 samples)."""
    from fish_diffusion_amd import RefineGANGenerator
    from oracle import refinegan_ref
    from tests.test_oracle_golden import _sine_noises
    g = load(f"refinegan_sine_{tag}")
    cfg = json.loads(str(g["config"]))
    sd = refinegan_ref.seeded_state(int(g["seed"]), cfg)
    assert sha1_state(sd) == str(g["weights_sha1"])
    gen = RefineGANGenerator(**cfg)
    assert "template_gen.merge.0.weight" in gen.state_dict()
    gen.load_folded_state(sd)
    gen = gen.to(dev).eval()
    B, _, T = g["mel"].shape
    noises = _sine_noises(g, cfg, B, T)
    wav = gen(g["mel"].to(dev), g["f0"].to(dev), noises=[nz.to(dev) for nz in noises])
    err = abs_err(wav.cpu(), g["wav"])
    print(f"refinegan sine {tag}: wav abs err {err:.3e}  (peak |wav| {float(g['wav'].abs().max()):.3f})")
    assert wav.shape == g["wav"].shape and err < WAV_ABS
    # torch-RNG mode draws rand(B, 1) first, like the reference: same stream => same waveform as the explicit draws
    torch.manual_seed(int(g["noise_seed"]))
    gen.rng = "torch"
    shapes = gen.noise_shapes(B, T)
    torch.rand(B, 1, device=dev)
    want = [torch.randn(s, device=dev) for s in shapes]
    torch.manual_seed(int(g["noise_seed"]))
    a = gen(g["mel"].to(dev), g["f0"].to(dev))
    b = gen(g["mel"].to(dev), g["f0"].to(dev), noises=want)
    assert torch.equal(a, b)
    gen.rng = "philox"
    c = gen(g["mel"].to(dev), g["f0"].to(dev))
    assert torch.isfinite(c).all() and float(c.abs().max()) <= 1.0


# ------------------------------------------------------------------------------------------------ exact-ragged batches
@pytest.mark.parametrize("net", ["small", "full"])
def test_exact_ragged_batches_equal_batch_one_runs_bit_for_bit(dev, net):
    """`GaussianDiffusion(..., lengths=)` / fdx_sampler_run_ragged: every member of a padded batch is computed exactly as if it ran
    alone at its own length (what tools/diffusion/inference.py:336-376 computes one segment at a time) -- BIT FOR BIT, for every
    sampler, for lengths straddling tile edges, and whatever the same buffers held beyond the lengths before (a previous, longer
    run).  Frames beyond an item's length are unspecified."""
    cfg = WN_SMALL if net == "small" else WN_FULL
    sd = wavenet_sd(cfg, 101 if net == "small" else 1234)
    diff = _diffusion(cfg, sd, dev)
    g = torch.Generator().manual_seed(77)
    cases = [([130, 64, 65, 1], 192), ([200, 113], 256)] if net == "small" else [([430, 300, 129], 448)]
    for lens, T in cases:
        B = len(lens)
        feats = torch.randn(B, T, 256, generator=g).to(dev)            # junk beyond the lengths on purpose
        x0 = torch.randn(B, 128, T, generator=g).to(dev)
        diff(feats, sampler_interval=250, x_init=x0)                    # leave full-length activations in every buffer
        for pred, iv in (("unipc", 100), ("plms", 100), ("naive", 100)):
            noise = torch.randn(1000 // iv, B, 128, T, generator=g).to(dev) if pred == "naive" else None
            got = diff(feats, sampler_interval=iv, noise_predictor=pred, x_init=x0, step_noise=noise, lengths=lens)
            again = diff(feats, sampler_interval=iv, noise_predictor=pred, x_init=x0, step_noise=noise, lengths=torch.tensor(lens))
            assert torch.equal(got, again)                               # (recorded graph replay)
            for b, n in enumerate(lens):
                alone = diff(feats[b:b + 1, :n].contiguous(), sampler_interval=iv, noise_predictor=pred, x_init=x0[b:b + 1, :, :n].contiguous(),
                             step_noise=None if noise is None else noise[:, b:b + 1, :, :n].contiguous())
                assert torch.equal(got[b, :n], alone[0]), (net, lens, pred, b)
    with pytest.raises(ValueError):
        diff(feats, lengths=lens, x_masks=torch.zeros(B, T, dtype=torch.bool, device=dev))
    with pytest.raises(ValueError):
        diff(feats, lengths=[T + 1] * B)
