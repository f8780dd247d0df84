import sys, time, torch
sys.path.insert(0, ".")
import bench
from fish_diffusion_amd import DENOISERS
from oracle import tfdec_ref
dev = torch.device("cuda:0")
mc = bench.TD_CFG
net = DENOISERS.build(dict(type="TransformerDecoderDenoiser", **mc))
net.load_state_dict(tfdec_ref.seeded_state(1, **mc)); net = net.to(dev).eval()
for B, T in ((1, 861), (8, 861)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, mc["mel_channels"], T, generator=g).to(dev); c = torch.randn(B, mc["condition_dim"], T, generator=g).to(dev)
    t = torch.full((B,), 500.0, device=dev)
    m = torch.zeros(B, T, dtype=torch.bool, device=dev); m[-1, T - 100:] = True
    for name, kw in (("no mask", {}), ("x_masks + cond_masks", dict(x_masks=m, cond_masks=m))):
        for _ in range(3): net(x, t, c, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): net(x, t, c, **kw)
        torch.cuda.synchronize(); print(f"B={B} T={T} {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per denoiser call")
