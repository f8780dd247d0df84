import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import WN_SMALL, wavenet_sd, rel_err
from tests.test_gpu_parity import _oracle_den, _diffusion
from oracle import sampler_ref
dev = torch.device("cuda", 0)
sd = wavenet_sd(WN_SMALL, 101)
den = _oracle_den(sd, WN_SMALL)
diff = _diffusion(WN_SMALL, sd, dev)
g = torch.Generator().manual_seed(40)
for T, lens, iv in ((128, [100, 90], 100), (128, [128, 90], 100), (128, [100, 90], 250), (100, [100, 90], 100), (128, [100, 90], 100)):
    fb = torch.zeros(2, T, 256)
    for b, n in enumerate(lens):
        fb[b, :n] = torch.randn(n, 256, generator=g)
    x = torch.randn(2, 128, T, generator=g)
    masks = torch.arange(T)[None] >= torch.tensor(lens)[:, None]
    mk = masks if masks.any() else None
    with torch.no_grad():
        ref = sampler_ref.diffusion_sample(den, fb, x_init=x, sampler_interval=iv, x_masks=mk, cond_masks=mk)
    out = diff(fb.to(dev), sampler_interval=iv, x_init=x.to(dev), x_masks=None if mk is None else mk.to(dev), cond_masks=None if mk is None else mk.to(dev)).cpu()
    print(T, lens, iv, [rel_err(out[b, :n], ref[b, :n]) for b, n in enumerate(lens)], "whole", rel_err(out, ref))
