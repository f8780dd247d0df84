#!/bin/bash
# round 5, GPU call 7: the instrumented residual-block kernels with the last shader-clock stamp and the real-time reading taken back to back
# (VERDICT r4 item 4a), and the headline line with board power sampled beside the clock
set -u
export TMPDIR=/tmp
out=gpurun_out/r05g; mkdir -p $out
timeout 300 python tools/ktrace.py 1 > $out/ktrace_headline_fp32.txt 2>&1
head -3 $out/ktrace_headline_fp32.txt | cut -c1-250; sed -n 4,12p $out/ktrace_headline_fp32.txt | cut -c1-250
timeout 300 python tools/ktrace.py 1 20 fp32 dense > $out/ktrace_headline_fp32_dense.txt 2>&1
sed -n 4,9p $out/ktrace_headline_fp32_dense.txt | cut -c1-250
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_headline.json 2> $out/bench_headline.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05g/bench_headline.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d.get("stages_ms"), d.get("clock_mhz"))
PY
