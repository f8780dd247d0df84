import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
diff, voc = bench.seeded_modules(dev)
B, T = 8, 896
feats, _ = bench.synth_inputs(B, T, dev, 1)
x0 = torch.randn(B, 128, T, device=dev)
def run(lens, n=3):
    for _ in range(2):
        diff(feats, sampler_interval=10, x_init=x0, lengths=lens)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        diff(feats, sampler_interval=10, x_init=x0, lengths=lens)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for lens in ([896] * 8, [448] * 8, [896, 896, 896, 896, 64, 64, 64, 64], [858, 813, 774, 755, 717, 672, 634, 554], None):
    print(lens, f"{run(lens):.1f} ms")
