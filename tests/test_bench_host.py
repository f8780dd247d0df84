"""CPU: the host-side arithmetic of bench.py (no GPU, no library calls)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_flops_match_the_survey():
    b = _bench()
    # 95.18 MFLOP per frame per step (incl. 0.02 T-independent), 1.2737 MFLOP per output sample (config_v1)
    assert abs(b.wavenet_flops_per_frame() / 1e6 - 95.16) < 0.05
    assert abs(b.nsf_flops_per_sample() / 1e6 - 1.2737) < 1e-3
    assert abs(b.nsf_flops_per_sample(b.NSF_V1_256) / 1e6 - 2.402) < 5e-3   # config_v1_256: 2.402 MFLOP per sample
    total = b.wavenet_flops_per_frame() * 861 * 100 + b.nsf_flops_per_sample() * 861 * 512
    assert abs(total / 1e12 - 8.755) < 0.01                               # one 10 s utterance, 100 steps
    # executed vs algorithmic (SURVEY 8d "never count hoisted work as achieved"): the 20 conditioner projections are
    # 10.49 MFLOP per frame and run once per utterance instead of once per step
    assert abs(b.wavenet_hoisted_flops_per_frame() / 1e6 - 10.49) < 0.01
    alg, exe = b.e2e_flops(861, 100, 861 * 512, 861)
    assert alg == total and abs((alg - exe) / 1e12 - 0.894) < 0.002       # 99 x 9.03 GFLOP
    a2, e2 = b.e2e_flops(0, 0, 32 * 1722 * 256, 0, b.NSF_V1_256, denoise=False)
    assert a2 == e2 and abs(a2 / 1e12 - 33.88) < 0.05                     # configs[2]: 32 x 1058.9 GFLOP


def test_config_aliases_cover_every_baseline_config():
    b = _bench()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert b.BASELINE_METRIC == base["metric"]
    assert {b.ALIASES[str(i)] for i in range(1, len(base["configs"]))} == {"headline", "vocoder", "sharded", "ddpm1000"}
    assert b.ALIASES["c2"] == "vocoder" and b.ALIASES["c3"] == "sharded" and b.ALIASES["c5"] == "ddpm1000"
    assert set(b.DEFAULT_STEPS) == set(b.ALIASES.values())
    # the default line carries every other BASELINE config and every SURVEY 8(f) row as a short run of its own
    assert set(b.EXTRA_CONFIGS) == {"vocoder", "sharded", "ddpm1000"} and set(b.EXTRA_WIDENING) == {"hifisinger_v2", "convnext", "tfdec"}
    assert set(b.EXTRA_CONFIGS) | set(b.EXTRA_WIDENING) | {"headline"} == set(b.ALIASES.values())


def test_widening_flop_formulas():
    """The SURVEY 8(f) rows' algorithmic work; tools/flops_reference.py checks the same formulas against torch.utils.flop_counter on the
    real reference modules (profiles/r04_flops_reference_check.json: ratio 1.0)."""
    b = _bench()
    assert abs(b.refinegan_flops(1722, b.RG_HIFISINGER) / 1e9 - 1246.98) < 0.01          # one 10 s item at hop 256, num_mels = 256
    assert b.refinegan_flops(1722, dict(b.RG_HIFISINGER, num_mels=128)) < b.refinegan_flops(1722, b.RG_HIFISINGER)
    cn, cn_h = b.convnext_flops_per_frame()
    assert abs(cn / 1e6 - 98.447) < 0.01 and abs(cn_h / 1e6 - 13.631) < 0.01
    td, td_h = b.tfdec_flops_per_frame(861)
    D, H, L = 512, 2048, 12
    assert td - td_h == 2.0 * (128 * H + H * D + L * (8 * D * D + 2 * D * H) + D * D + D * 128) + L * 2 * 4.0 * 861 * D
    assert b.hifisinger_frontend_flops(1722) == 2.0 * 1722 * (768 * 256 + 2 * 256 * 256)
    ref = os.path.join(ROOT, "profiles", "r04_flops_reference_check.json")
    if os.path.exists(ref):
        with open(ref) as f:
            rows = json.load(f)["rows"]
        for k, v in rows.items():
            if isinstance(v, dict):
                assert abs(v.get("ratio", v.get("ratio_without_attention_products")) - 1.0) < 1e-9, k


def test_usable_cores_and_traffic_file():
    b = _bench()
    n = b.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    traffic, src = b.pmc_traffic("headline", "convgate", {"config": "headline", "batch": 1, "frames": 861})
    assert src is None or os.path.exists(os.path.join(ROOT, src))
    if traffic is not None:
        with open(os.path.join(ROOT, src)) as f:
            d = json.load(f)
        k = next(v for name, v in d["kernels"].items() if "EpiGate" in name)
        assert traffic == k["fetch_bytes"] + k["write_bytes"]
        assert abs(k["fetch_bytes"] - 2 * k["FETCH_SIZE"] * 1024) < 2048    # FETCH_SIZE is stored rounded to 0.1 KiB
    # counters are only valid for the workload they were taken on
    assert b.pmc_traffic("headline", "convgate", {"config": "headline", "batch": 4, "frames": 861}) == (None, None)


def test_make_batches_rules():
    from fish_diffusion_amd.pipeline import make_batches
    assert make_batches([10, 9, 8, 3], 2) == [[0, 1], [2], [3]]              # pairing 8 with 3 would pad 3 -> 8: cheaper apart
    assert make_batches([10, 9, 8, 3], 2, max_pad_ratio=0.25) == [[0, 1], [2], [3]]
    assert make_batches([861, 850, 800, 790, 740, 700, 690, 640], 8) == [[0, 1, 2, 3, 4, 5, 6, 7]]   # a full batch beats [6, 2]
    assert make_batches([861, 800, 700, 650, 600, 560, 540, 520], 8) == [[0, 1], [2, 3], [4, 5, 6, 7]]   # ... unless the spread is wide
    assert make_batches([861, 800, 700, 650, 600, 560, 540, 520], 8, max_pad_ratio=0.25) == [[0, 1], [2, 3], [4, 5, 6, 7]]   # no member padded > 25 %
    assert make_batches([5, 5, 5], 8) == [[0, 1, 2]]
    assert make_batches([], 4) == []
    flat = sorted(i for b in make_batches([7, 100, 90, 95, 20, 21, 60], 3) for i in b)
    assert flat == list(range(7))


def _last_json(stdout: str):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_bench_gpus2_spawns_two_ranks_by_itself():
    """`python bench.py --gpus N` (no wrapper, WORLD_SIZE unset) must start N ranks.  --dry-run runs the
    launcher, the rendezvous, the arena broadcast, the barrier-bracketed timing and the per-rank gather on CPU ranks over gloo."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1",
                        "--config", "sharded"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert r.stdout.strip().splitlines()[-1].startswith("{")          # rank 0's JSON line is the last line of stdout
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["backend"] == "gloo"
    assert d["launched_by"] == "fish_diffusion_amd.dist.launch_ranks"
    assert len(d["per_rank_ms"]) == 2 and len(d["per_rank_frames"]) == 2
    assert sum(d["per_rank_utterances"]) == 64 and d["scaling"] == "strong"      # the 64 utterances are a partition over the two ranks
    # without the GPUs it asks for, the real run refuses instead of degrading to fewer ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "refusing to run fewer ranks" in (r.stderr + r.stdout)


def test_bench_under_torch_distributed_run_is_a_rank_not_a_launcher():
    """The driver's own command line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) must not spawn again."""
    import subprocess
    import sys
    from fish_diffusion_amd.dist import free_port
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["launched_by"] == "external launcher" and d["scaling"] == "weak"


def test_launch_ranks_propagates_a_failing_rank_and_stops_the_rest():
    import sys
    import time
    from fish_diffusion_amd.dist import launch_ranks
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(60)\n"
    t0 = time.monotonic()
    rc = launch_ranks(3, [sys.executable, "-c", code], need_gpus=False)
    assert rc == 7 and time.monotonic() - t0 < 30          # ranks 0 and 2 were stopped, not waited for
    assert launch_ranks(2, [sys.executable, "-c", "import os; assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'"],
                        need_gpus=False) == 0
