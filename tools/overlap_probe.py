"""Experiment (round 6): does the vocoder pass of utterance i hide under the sampler of utterance i + 1 when the two run on two HIP streams?
(round 2 measured +1.4 % TIME for "sampler || vocoder on two streams"; the vocoder is 2.3x faster since.)  Same work, same results either way.
    python tools/overlap_probe.py [steps]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from benchkit.workloads import build_work

def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    args = SimpleNamespace(storage="fp32", seconds=10.0, batch=None, interval=None, prof_stride=None, virtual_world=8, no_exact=False)
    w = build_work("headline", args, dev, 0, 1, steps + 2)
    diff, voc, pool, f0, iv = w.diff, w.voc, w.pool, w.f0, w.interval
    side = torch.cuda.Stream(device=dev)

    def serial(k):
        mel = diff(pool[k % len(pool)], sampler_interval=iv)
        return voc.model(mel.transpose(1, 2), f0, mel_scale=2.30259)

    def overlapped(k):
        cur = torch.cuda.current_stream(dev)
        mel = diff(pool[k % len(pool)], sampler_interval=iv)
        melT = mel.transpose(1, 2).contiguous()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            wav = voc.model(melT, f0, mel_scale=2.30259)
        melT.record_stream(side)
        return wav

    res = {}
    for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial again", serial), ("overlapped again", overlapped)):
        for k in range(2):
            fn(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [fn(2 + k) for k in range(steps)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res[name] = dt
        print(f"{name:18s} {dt * 1e3:8.3f} ms per utterance   {w.audio_s / dt:7.2f} x real-time", flush=True)
    a = serial(5); torch.cuda.synchronize(); b = overlapped(5); torch.cuda.synchronize()
    print("same waveform bits:", torch.equal(a, b))

if __name__ == "__main__":
    main()
