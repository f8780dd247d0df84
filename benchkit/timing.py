"""Measurement for bench.py: the timed region (barrier + synchronize on both sides, MAX over ranks), the dominant kernel's launch-stream
events (`fdx_prof_*`), the committed PMC traffic files, the shader-clock / board-power sampler."""
from __future__ import annotations

import ctypes as C
import glob
import json
import os
import threading
import time
from types import SimpleNamespace

import torch

from .flops import PEAK_HBM_GBS, WN_CFG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


TRAFFIC_KEYS = {"convgate": ("EpiGate",), "outproj": ("EpiResSkip",), "nsf_resblock": ("2, false, 1, EpiResblock",),
                "rg_resblock": ("2, false, 1, EpiResblock",), "cn_pwconv1": ("EpiBiasAct16S",), "td_attn": ("k_attn_qs",)}


def pmc_traffic(config: str, kernel: str, expect: dict):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_traffic.py -> profiles/*_pmc_traffic.json;
    FETCH_SIZE and WRITE_SIZE need separate passes, so bench.py cannot collect them itself).  A file is only used for the
    workload it was collected on: its "workload" record must equal `expect` (files without one are the round-1 headline files:
    batch 1, T = 861)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        wl = d.get("workload", {"config": "headline", "batch": 1, "frames": 861})
        if wl != expect:
            continue
        for k, v in d["kernels"].items():
            if any(s in k for s in TRAFFIC_KEYS[kernel]):
                return v["hbm_bytes"], os.path.relpath(path, ROOT)
    return None, None


def prof_begin(handle, kind, stride):
    from fish_diffusion_amd import _lib
    _lib.check(_lib.lib().fdx_prof_select(handle.h, kind), handle.h)
    _lib.check(_lib.lib().fdx_prof_enable(handle.h, stride), handle.h)


def prof_pause(handle):
    from fish_diffusion_amd import _lib
    _lib.check(_lib.lib().fdx_prof_enable(handle.h, -1), handle.h)


def prof_end(handle):
    """(launches, avg_ms, flops_per_launch, label) of the launches recorded since prof_begin.  `label` is the library's own
    description of the kernel instantiation those launches ran (fdx_prof_label) -- never a literal in this file."""
    from fish_diffusion_amd import _lib
    n, ms, fl = C.c_int(), C.c_double(), C.c_double()
    buf = C.create_string_buffer(320)
    _lib.check(_lib.lib().fdx_prof_label(handle.h, buf, len(buf)), handle.h)
    _lib.check(_lib.lib().fdx_prof_read(handle.h, C.byref(n), C.byref(ms), C.byref(fl)), handle.h)
    _lib.check(_lib.lib().fdx_prof_enable(handle.h, 0), handle.h)
    if not n.value:
        return 0, 0.0, 0.0, ""
    return n.value, ms.value / n.value, fl.value, buf.value.decode()


def graph_stats(handle):
    """(recordings, replays, shapes held) of the handle's sampler-graph cache (fdx_graph_stats)."""
    from fish_diffusion_amd import _lib
    cap, lau, cached = C.c_long(), C.c_long(), C.c_int()
    _lib.check(_lib.lib().fdx_graph_stats(handle.h, C.byref(cap), C.byref(lau), C.byref(cached)), handle.h)
    return {"recorded": cap.value, "replayed": lau.value, "held": cached.value}


def first_call_entry(m, w):
    """`first_call_ms` of a line: what the FIRST step of the shape cost against the steady step."""
    if m.first_ms is None:
        return None
    steady = m.dt / m.steps * 1e3
    return {"first_step_ms": round(m.first_ms, 2), "steady_step_ms": round(steady, 3), "cold_cost_ms": round(m.first_ms - steady, 2),
            "sampler_graphs": m.graph,
            "note": "first step of this row shape in the process (weights already packed: see weights_pack_*): conditioner hoisting, workspace allocation, "
                    "recording + instantiating the sampler body as a hipGraph, one run; later steps of the shape replay the recording"
                    + ("" if not w.warm else "; this config's first step is its 10-step warm-up pass")}


def roofline_entry(kernel_desc, n, avg_ms, flops, peak, sampling, traffic=None, traffic_src=None, alg_bytes=None):
    ach = flops / (avg_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": kernel_desc, "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)",
            "traffic_source": traffic_src, "algorithmic_bytes": alg_bytes,
            "hbm_fraction": (round(traffic / (avg_ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4) if traffic else None),
            "peak_note": "nominal fp32 matrix peak at the 2.4 GHz boost clock.  tools/ubench/mfmaclk.hip on this part (profiles/r04_mfma_clock_ubench.txt): an "
                         "MFMA-only v_mfma_f32_16x16x4_f32 loop on 256 CUs sustains 155.1 TFLOP/s at 2.39 GHz, as one long launch and as a chain of 12 / 25 us "
                         "launches alike; with the residual-block K loop's load mix (6 dwordx4 per 16 MFMAs from L2) 116.5 at the SAME 2.39 GHz: operand delivery "
                         "bounds the K loop.  The library's own kernels, same counters (s_memtime / s_memrealtime per wave, instrumented build, last two stamps "
                         "taken back to back: profiles/r05_ktrace_headline_fp32_adjacent_stamps.txt), read 2.10 (conv + gate) / 2.18 GHz (out-projection) while sclk "
                         "reports 2.38-2.40 at ~1100 W of board power (`clock_mhz`).  Round 5 ruled out wait states (a wave that only sleeps reads 2.397 GHz), barriers, "
                         "LDS reductions, exp phases, cold-load waits, combined L2 + LDS + MFMA load (all 2.38-2.39, profiles/r05_clock_*_ubench.txt) and the stamps "
                         "themselves; no cause is named (profiles/NOTES.md round 5 item 4) -- `peak` stays the nominal 157.3",
            "launches_timed": n, "sampling": sampling, "avg_launch_us": round(avg_ms * 1e3, 2),
            "timing": "hipExtLaunchKernel start/stop events on the launch stream", "flops_per_launch": flops}


class SclkSampler:
    """Shader clock of THIS GPU as the driver reports it (sysfs pp_dpm_sclk: the level marked '*'), sampled from a thread while the
    timed region runs.  The card is matched by PCI address (torch's device properties); no match -> no samples (reported as such)."""

    def __init__(self, dev_index: int, period_s: float = 0.02):
        self.period, self.samples, self._stop, self._thr, self.path, self.why = period_s, [], threading.Event(), None, None, None
        self.power_path, self.power = None, []      # board power (hwmon, microwatts) beside the clock: VERDICT r4 item 4(a)
        try:
            p = torch.cuda.get_device_properties(dev_index)
            want = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            for card in sorted(glob.glob("/sys/class/drm/card*/device")):
                if os.path.basename(os.path.realpath(card)) == want and os.path.exists(os.path.join(card, "pp_dpm_sclk")):
                    self.path = os.path.join(card, "pp_dpm_sclk")
                    pw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input")))
                    self.power_path = pw[0] if pw else None
            if self.path is None:
                self.why = f"no /sys/class/drm/card*/device matches PCI {want}"
        except Exception as e:   # noqa: BLE001
            self.why = f"{type(e).__name__}: {e}"

    def _read(self):
        try:
            for ln in open(self.path).read().splitlines():
                if ln.rstrip().endswith("*"):
                    return float(ln.split(":")[1].strip().split("M")[0])
        except Exception:
            return None
        return None

    def _run(self):
        while not self._stop.is_set():
            v = self._read()
            if v is not None:
                self.samples.append(v)
            if self.power_path:
                try:
                    self.power.append(float(open(self.power_path).read()) / 1e6)
                except Exception:   # noqa: BLE001
                    pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self.path:
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thr:
            self._thr.join(timeout=1.0)

    def report(self):
        if not self.samples:
            return {"mean": None, "samples": 0, "source": self.path, "note": self.why or "no samples"}
        s = sorted(self.samples)
        out = {"mean": round(sum(s) / len(s), 1), "median": s[len(s) // 2], "min": s[0], "max": s[-1], "samples": len(s),
               "period_ms": self.period * 1e3, "source": self.path,
               "note": "sysfs pp_dpm_sclk ('*' level) of this GPU, sampled by a host thread during the timed region"}
        if self.power:
            out["board_power_w"] = {"mean": round(sum(self.power) / len(self.power), 1), "max": round(max(self.power), 1), "samples": len(self.power),
                                    "source": self.power_path}
        return out



def measure(w, steps, warmup, args, dev, do_prof, sclk=False, prof_outside=False):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by synchronize + barrier + synchronize on both sides; MAX over ranks.
    The dominant kernel is timed (launch-stream events) on the FIRST timed step only -- it needs the eager launch path; the other steps
    replay the recorded hipGraph.  `prof_outside` (the widening rows whose denoiser call is ~150 launches of 5-25 us: an eager step is bound by
    the host's launch rate, 250 ms against 205 for the transformer, and would be half of a 2-step timed region): the kernel is timed on ONE
    EXTRA step after the timed region instead, and every timed step replays the graph."""
    from fish_diffusion_amd import dist as fdist
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1

    def sync_barrier():
        torch.cuda.synchronize()
        t_local = time.perf_counter()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()
        return t_local

    # the first call of a shape: conditioner hoisting, buffer allocation, the sampler graph's RECORDING (4 k nodes for the WaveNet, ~13 k for the
    # transformer) and one run -- what a serving loop pays per new row shape (the LRU holds 48 recorded shapes); weights are packed before this
    first_ms = None
    for k in range(warmup):
        if k == 0:
            torch.cuda.synchronize()
            t_first = time.perf_counter()
        (w.warm or w.step)(k)
        if k == 0:
            torch.cuda.synchronize()
            first_ms = (time.perf_counter() - t_first) * 1e3
    sync_barrier()
    prof_in = do_prof and not prof_outside
    if prof_in:
        prof_begin(w.prof_handle(), w.prof_kind, w.stride)
        sync_barrier()
    sampler = SclkSampler(dev.index or 0) if sclk else None
    if sampler:
        sampler.__enter__()
    t0 = time.perf_counter()
    for k in range(steps):
        out = w.step(warmup + k)
        if k == 0 and prof_in:
            prof_pause(w.prof_handle())
    t_local = sync_barrier()
    dt = time.perf_counter() - t0
    if sampler:
        sampler.__exit__(None, None, None)
    per_rank = fdist.gather_stats([(t_local - t0) / steps * 1e3, w.audio_s, float(w.cfg_extra.get("frames_this_rank", w.B * w.T)),
                                   float(w.cfg_extra.get("utterances_this_rank", w.B))], dev)   # [world, 4]
    dt = fdist.barrier_max(dt, dev)
    if getattr(w, "failures", None) is not None:      # which utterances the job lost, over all ranks (none, on synthetic input)
        w.cfg_extra["failed_utterances"] = fdist.gather_failed(sorted({i for i, _ in w.failures}), dev)
    del out
    roofline = None
    if do_prof:
        if prof_outside:
            prof_begin(w.prof_handle(), w.prof_kind, w.stride)
            w.step(warmup + steps)
            torch.cuda.synchronize()
        n, avg_ms, fl, label = prof_end(w.prof_handle())
        if n:
            traffic, traffic_src = pmc_traffic(w.name, w.traffic_key, w.traffic_expect)
            where = "one extra step after the timed region" if prof_outside else "the first timed step"
            roofline = roofline_entry(f"{label}: {w.kwhat}", n, avg_ms, fl, w.peak, f"every {w.stride}th launch of {where}", traffic, traffic_src,
                                      w.alg_bytes)
    audio_all = fdist.sum_over_ranks(w.audio_s, dev)      # weak configs: world x audio_s; sharded: the ranks' shards differ
    alg, exe = fdist.sum_over_ranks(w.alg, dev) / world, fdist.sum_over_ranks(w.exe, dev) / world   # per-GPU means
    graph = None
    try:
        graph = graph_stats(w.prof_handle())
    except Exception:   # noqa: BLE001  (a workload without a sampler handle)
        pass
    return SimpleNamespace(dt=dt, steps=steps, warmup=warmup, per_rank=per_rank, roofline=roofline, value=steps * audio_all / dt, alg=alg, exe=exe,
                           first_ms=first_ms, graph=graph,
                           e2e_alg=alg * steps / dt / 1e12, e2e_exe=exe * steps / dt / 1e12, clock=sampler.report() if sampler else None, world=world)


def other_kernel(w, kind, args, steps_done):
    """The second residual-block kernel, timed on one extra step outside the timed region."""
    prof_begin(w.prof_handle(), kind, w.stride)
    w.step(steps_done)
    torch.cuda.synchronize()
    n, avg_ms, fl, label = prof_end(w.prof_handle())
    if not n:
        return None
    tr, src = pmc_traffic(w.name, "outproj", w.traffic_expect)
    C_, M_ = WN_CFG["residual_channels"], fl / (2.0 * 2 * WN_CFG["residual_channels"] ** 2)   # columns per launch, from its flops
    esz = 2 if w.bf16 else 4
    # weights [2C x C] + Z in + X in/out + SK in/out + next layer's Y out (fp32 residual stream in every mode)
    ob = esz * (2 * C_ * C_ + C_ * M_) + 4 * (4 * C_ * M_) + esz * C_ * M_
    e = roofline_entry(f"{label}: 1x1 out-projection + residual / skip epilogue" + (" (HBM-bound: the fp32 residual stream and skip sum "
                       "are read and written every layer)" if w.bf16 else ""), n,
                       avg_ms, fl, w.peak, f"every {w.stride}th launch of one extra step outside the timed region", tr, src, int(ob))
    if w.bf16:
        e["bound"] = "hbm"
    return e
