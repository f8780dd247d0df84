import os, time, torch, sys
sys.path.insert(0, '.')
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)' ; free -g | head -2")
from oracle import wavenet_ref
cfg = dict(mel_channels=128, d_encoder=256, residual_channels=512, residual_layers=20, use_linear_bias=True)
sd = wavenet_ref.seeded_wavenet_state(1, **cfg)
x = torch.randn(1,128,861); c = torch.randn(1,256,861)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    with torch.no_grad():
        wavenet_ref.wavenet_forward(sd, x, torch.tensor([5.]), c, None, None, residual_layers=20, dilation_cycle=4)
        t0=time.perf_counter()
        for _ in range(2): wavenet_ref.wavenet_forward(sd, x, torch.tensor([5.]), c, None, None, residual_layers=20, dilation_cycle=4)
        print("threads", nt, "wavenet fwd s", (time.perf_counter()-t0)/2, flush=True)
