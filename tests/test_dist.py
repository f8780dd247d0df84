"""CPU, world_size 2, gloo: the multi-GPU layer (fish_diffusion_amd/dist.py).  On the GPU box the same code runs
over "nccl" (= RCCL); the arenas are plain byte tensors, so the broadcast logic is identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from fish_diffusion_amd import _lib, dist as fdist
    from fish_diffusion_amd.nsf_hifigan import Generator
    from fish_diffusion_amd.wavenet import WaveNet
    r, lr, w = fdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    cpu = torch.device("cpu")
    # different random weights on every rank: after the broadcast everyone must hold rank 0's packed bytes
    torch.manual_seed(100 + rank)
    net = WaveNet(mel_channels=128, d_encoder=256, residual_channels=64, residual_layers=3, dilation_cycle=2, use_linear_bias=True)
    arena = fdist.broadcast_arena(net._desc, "wavenet", net._params() if rank == 0 else None, cpu)
    h = dict(resblock="1", upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=64,
             resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3], [1, 3]], num_mels=128, hop_size=8, sampling_rate=44100)
    gen = Generator(h)
    with torch.no_grad():
        garena = fdist.broadcast_arena(gen._desc, "nsf", gen.folded_weights() if rank == 0 else None, cpu)
    if rank == 0:
        local = torch.from_numpy(_lib.pack_on_host(net._desc, net._params(), "wavenet").view(np.uint8))
        assert torch.equal(local, arena)
    # the other denoisers ride the same broadcast (dist.broadcast_model_weights keys on the module's `_KIND`)
    from fish_diffusion_amd import ConvNext, TransformerDecoderDenoiser
    for j, den in enumerate((ConvNext(mel_channels=16, dim=64, mlp_factor=2, condition_dim=24, num_layers=2),
                             TransformerDecoderDenoiser(mel_channels=16, dim=128, mlp_factor=1, condition_dim=24, num_layers=1))):
        a = fdist.broadcast_arena(den._desc, den._KIND, den._params() if rank == 0 else None, cpu)
        np.save(os.path.join(out_dir, f"darena{j}_{rank}.npy"), a.numpy())
    np.save(os.path.join(out_dir, f"arena{rank}.npy"), arena.numpy())
    np.save(os.path.join(out_dir, f"garena{rank}.npy"), garena.numpy())
    # per-rank stats gather + max-over-ranks timing
    stats = fdist.gather_stats([rank + 1.0, 10.0 * rank], cpu)
    assert stats.shape == (world, 2) and stats[:, 0].tolist() == [1.0, 2.0]
    assert fdist.barrier_max(0.5 + rank, cpu) == 1.5
    # utterance sharding is a partition
    lengths = [516, 861, 700, 861, 600, 530, 800]
    mine = fdist.shard_utterances(lengths, rank, world)
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), np.array(mine))
    tdist.barrier()
    tdist.destroy_process_group()


def test_broadcast_and_sharding_world2(tmp_path, lib_built):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a0, a1 = np.load(tmp_path / "arena0.npy"), np.load(tmp_path / "arena1.npy")
    assert a0.size > 0 and np.array_equal(a0, a1)
    g0, g1 = np.load(tmp_path / "garena0.npy"), np.load(tmp_path / "garena1.npy")
    assert g0.size > 0 and np.array_equal(g0, g1)
    for j in range(2):
        d0, d1 = np.load(tmp_path / f"darena{j}_0.npy"), np.load(tmp_path / f"darena{j}_1.npy")
        assert d0.size > 0 and np.array_equal(d0, d1)
    s0, s1 = np.load(tmp_path / "shard0.npy").tolist(), np.load(tmp_path / "shard1.npy").tolist()
    assert sorted(s0 + s1) == list(range(7)) and not set(s0) & set(s1)
    lengths = [516, 861, 700, 861, 600, 530, 800]
    assert s0[0] == 1 and s1[0] == 3                      # longest first, ties in original order
    assert abs(sum(lengths[i] for i in s0) - sum(lengths[i] for i in s1)) <= max(lengths)


def test_shard_utterances_single_rank_is_sorted_identity():
    from fish_diffusion_amd.dist import shard_utterances
    assert shard_utterances([3, 9, 5], 0, 1) == [1, 2, 0]
    assert shard_utterances([], 0, 4) == []


# ------------------------------------------------------------------------------------------------ failure isolation (round 5)
class _FakeDiffusion:
    """Host stand-in with GaussianDiffusion's call contract: mel = mean of the features per frame; an utterance whose features hold a NaN
    makes the WHOLE micro-batch raise (what a device-side failure does to a batch)."""
    mel_bins = 4

    def __init__(self):
        self.calls = []

    def __call__(self, feat, sampler_interval=None, noise_predictor=None, **kw):
        self.calls.append(feat.shape[0])
        if torch.isnan(feat).any():
            raise RuntimeError("FDX_E_HIP: poisoned micro-batch")
        return feat.mean(-1, keepdim=True).expand(-1, -1, self.mel_bins).contiguous()


class _FakeGenerator:
    h = {"hop_size": 2}

    def __call__(self, mel, f0, mel_scale=1.0, **kw):
        return (mel.mean(1, keepdim=True).repeat_interleave(2, dim=-1) + f0.repeat_interleave(2, dim=-1)[:, None]) * mel_scale


def _utterances(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    lens = [int(v) for v in torch.randint(5, 40, (n,), generator=g)]
    return [torch.randn(t, 6, generator=g) for t in lens], [torch.rand(t, generator=g) for t in lens]


def test_synthesize_isolates_failing_utterances():
    """SURVEY section 5 / tools/preprocessing/extract_features.py:175-217 (`safe_process`): one bad utterance must not lose the batch."""
    from fish_diffusion_amd import pipeline
    feats, f0s = _utterances(11)
    feats[3][2, 1] = float("nan")          # fails inside the sampler: its batch raises, is re-run member by member
    f0s[7] = f0s[7][:-1]                   # fails validation (length mismatch): never reaches a batch
    feats[9] = torch.zeros(0, 6)           # empty
    diff, gen = _FakeDiffusion(), _FakeGenerator()
    with pytest.raises((RuntimeError, ValueError, IndexError)):
        pipeline.synthesize(diff, gen, feats, f0s, max_batch=4, exact=False)
    fails = []
    res = pipeline.synthesize(diff, gen, feats, f0s, max_batch=4, exact=False, on_error="isolate", failures=fails)
    assert sorted(i for i, _ in fails) == [3, 7, 9]
    assert sorted(i for i, _, _ in res) == [0, 1, 2, 4, 5, 6, 8, 10]
    assert "poisoned" in dict(fails)[3] and "f0 must be" in dict(fails)[7]
    # the survivors are what a clean run of them alone gives
    clean = {i: w for i, _, w in pipeline.synthesize(_FakeDiffusion(), gen, [feats[i] for i in (0, 1, 2, 4, 5, 6, 8, 10)],
                                                     [f0s[i] for i in (0, 1, 2, 4, 5, 6, 8, 10)], max_batch=1, exact=False)}
    for k, i in enumerate((0, 1, 2, 4, 5, 6, 8, 10)):
        got = next(w for j, _, w in res if j == i)
        assert torch.allclose(got, clean[k], atol=1e-6)


def _worker_failures(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from fish_diffusion_amd import dist as fdist, pipeline
    fdist.init_process_group("gloo")
    feats, f0s = _utterances(23, seed=5)
    feats[4][0, 0] = float("nan")
    feats[17][1, 1] = float("nan")
    f0s[10] = f0s[10][:3]
    fails = []
    res = pipeline.synthesize(_FakeDiffusion(), _FakeGenerator(), feats, f0s, max_batch=3, exact=False, rank=rank, world=world,
                              on_error="isolate", failures=fails)
    everyone = fdist.gather_failed([i for i, _ in fails], torch.device("cpu"))
    done = fdist.gather_failed([i for i, _, _ in res], torch.device("cpu"))      # the same two collectives carry the finished ids
    np.save(os.path.join(out_dir, f"failed{rank}.npy"), np.array(everyone))
    np.save(os.path.join(out_dir, f"done{rank}.npy"), np.array(done))
    np.save(os.path.join(out_dir, f"local{rank}.npy"), np.array([i for i, _ in fails], dtype=np.int64))
    tdist.barrier()
    tdist.destroy_process_group()


def test_failed_ids_are_gathered_world4(tmp_path, lib_built):
    """world 4 over gloo, failures on some ranks and none on others: every rank ends with the same list of lost utterances, and
    lost + done is a partition of the job."""
    port = _free_port()
    mp.spawn(_worker_failures, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    failed = [np.load(tmp_path / f"failed{r}.npy").tolist() for r in range(4)]
    done = [np.load(tmp_path / f"done{r}.npy").tolist() for r in range(4)]
    local = [np.load(tmp_path / f"local{r}.npy").tolist() for r in range(4)]
    assert all(f == [4, 10, 17] for f in failed)
    assert all(d == done[0] for d in done) and sorted(done[0] + failed[0]) == list(range(23))
    assert sorted(sum(local, [])) == [4, 10, 17] and any(len(x) == 0 for x in local)
