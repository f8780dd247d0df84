"""CPU, world_size 2, gloo: the multi-GPU layer (fish_diffusion_amd/dist.py).  On the GPU box the same code runs
over "nccl" (= RCCL); the arenas are plain byte tensors, so the broadcast logic is identical."""
import os
import socket

import numpy as np
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
