#!/bin/bash
# round 5, GPU call 15: the 2-D tile -> XCD map threshold of the residual-block kernels on the sharded geometry (1664 workgroups: below the 2048 default)
set -u
export TMPDIR=/tmp
out=gpurun_out/r05o; mkdir -p $out
for thr in 2048 1024; do
  FDX_XCD_RECT=$thr timeout 200 python bench.py --config sharded --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-extras > $out/sharded_rect$thr.json 2> $out/sharded_rect$thr.err
  python - <<PY
import json
d = json.loads(open("$out/sharded_rect$thr.json").read().strip().splitlines()[-1])
print("FDX_XCD_RECT=$thr sharded", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], [o.get("avg_launch_us") for o in (d.get("other_kernels") or {}).values()] if isinstance(d.get("other_kernels"), dict) else d.get("other_kernels"))
PY
done
for c in FETCH_SIZE WRITE_SIZE; do
  FDX_XCD_RECT=1024 timeout 150 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_sh_$c -o pmc -- python tools/pmc_mini.py sharded > $out/sharded_pmc_$c.log 2>&1
done
python tools/pmc_traffic.py /tmp/pmc_sh_FETCH_SIZE/pmc_results.db /tmp/pmc_sh_WRITE_SIZE/pmc_results.db sharded > $out/sharded_rect1024_pmc_traffic.json 2> $out/err.txt
python - <<'PY'
import json
t = json.load(open("gpurun_out/r05o/sharded_rect1024_pmc_traffic.json"))
for k, v in t["kernels"].items(): print("rect1024", k[:80], v["launches"], round(v["hbm_bytes"] / 1e6, 2), "MB")
PY
