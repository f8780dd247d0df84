#!/bin/bash
# round 5: the widening denoisers at batch 8 (the exact-ragged serving regime) as bench lines
set -u
out=gpurun_out/r05r; mkdir -p $out
for cfg in tfdec convnext; do
  timeout 300 python bench.py --config $cfg --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-extras > $out/${cfg}_b8.json 2> $out/${cfg}_b8.err
  python - <<PY
import json
d = json.loads(open("$out/${cfg}_b8.json").read().strip().splitlines()[-1])
print("$cfg batch 8:", d["value"], "audio-s/s,", d["ms_per_step"], "ms per step, e2e", d["end_to_end"]["frac_of_peak"], "kernel", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
done
