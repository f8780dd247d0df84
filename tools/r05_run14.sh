#!/bin/bash
# round 5, GPU call 14: PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate) of the transformer and sharded geometries on a SHORT eager run
set -u
export TMPDIR=/tmp
out=gpurun_out/r05n; mkdir -p $out
for cfg in tfdec sharded; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 150 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${cfg}_$c -o pmc -- python tools/pmc_mini.py $cfg > $out/${cfg}_pmc_$c.log 2>&1
    tail -1 $out/${cfg}_pmc_$c.log | cut -c1-120
  done
  python tools/pmc_traffic.py /tmp/pmc_${cfg}_FETCH_SIZE/pmc_results.db /tmp/pmc_${cfg}_WRITE_SIZE/pmc_results.db $cfg > $out/${cfg}_pmc_traffic.json 2> $out/${cfg}_pmc_traffic.err
  python - <<PY
import json
try:
    t = json.load(open("$out/${cfg}_pmc_traffic.json"))
    for k, v in t["kernels"].items():
        print("$cfg", k[:80], v["launches"], round(v["hbm_bytes"] / 1e6, 2), "MB")
except Exception as e:
    print("$cfg", "failed:", e, open("$out/${cfg}_pmc_traffic.err").read()[-300:])
PY
done
