#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are collected
in SEPARATE runs: they do not fit one pass on gfx950's 4 TCC counters).

    python tools/pmc_traffic.py <fetch_db> <write_db> [config] > profiles/rNN_<config>_pmc_traffic.json

`config` (headline | vocoder | sharded | ddpm1000 | ddpm1000_bf16, default headline) is recorded as the file's "workload" -- the
bench.py configuration the passes were collected on; bench.py only quotes a file's numbers for that workload.

Units and corrections, per /opt/skills/guides/MI355X_MICROARCH.md section HBM: both counters are in KiB-like units of
1024 B as reported by rocprofv3; on gfx950 FETCH_SIZE counts 128-byte requests at 64 B, i.e. reports half of the bytes
of wide coalesced reads -> fetch_bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is used as reported (it matches the
kernel's algorithmic store volume exactly: 1728 KiB = 512 x 864 x 4 B for the gate kernel)."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name",
                     (counter,)).fetchall()
    return {k: (n, v) for k, n, v in rows}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    cfg = sys.argv[3] if len(sys.argv) > 3 else "headline"
    workloads = {"headline": {"config": "headline", "batch": 1, "frames": 861}, "vocoder": {"config": "vocoder", "batch": 32, "frames": 1722},
                 "sharded": {"config": "sharded", "batch": 8, "frames": 858},   # rank 0's micro-batch of 8: longest member 858 frames (bench.py's seeded lengths)
                 "ddpm1000": {"config": "ddpm1000", "batch": 16, "frames": 861},
                 "ddpm1000_bf16": {"config": "ddpm1000_bf16", "batch": 16, "frames": 861},
                 "ddpm1000_fp16x3": {"config": "ddpm1000_fp16x3", "batch": 16, "frames": 861},
                 "headline_fp16x3": {"config": "headline_fp16x3", "batch": 1, "frames": 861},
                 "hifisinger_v2": {"config": "hifisinger_v2", "batch": 16, "frames": 1722},
                 "convnext": {"config": "convnext", "batch": 1, "frames": 861}, "tfdec": {"config": "tfdec", "batch": 1, "frames": 861}}
    out = {"source": {"fetch_db": sys.argv[1], "write_db": sys.argv[2]}, "workload": workloads[cfg],
           "note": "bytes per launch; fetch = 2 x FETCH_SIZE x 1024 (gfx950 half-count correction), write = WRITE_SIZE x 1024",
           "kernels": {}}
    for k in fetch:
        if not any(t in k for t in ("convgemm", "bf16lds", "f16s64", "k_attn", "k_resblock1_fused", "k_dwconv_stats")):
            continue
        n, f = fetch[k]
        w = write.get(k, (0, 0.0))[1]
        out["kernels"][k.replace("void ", "").replace("fdx::", "")] = {
            "launches": n, "FETCH_SIZE": round(f, 1), "WRITE_SIZE": round(w, 1),
            "fetch_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024), "hbm_bytes": int(2 * f * 1024 + w * 1024)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
