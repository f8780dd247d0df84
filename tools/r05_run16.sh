#!/bin/bash
set -u
export TMPDIR=/tmp
out=gpurun_out/r05p; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_sh_$c -o pmc -- python tools/pmc_mini.py sharded > $out/sharded_pmc_$c.log 2>&1
done
python tools/pmc_traffic.py /tmp/pmc_sh_FETCH_SIZE/pmc_results.db /tmp/pmc_sh_WRITE_SIZE/pmc_results.db sharded > $out/sharded_pmc_traffic.json 2> $out/err.txt
python - <<'PY'
import json
t = json.load(open("gpurun_out/r05p/sharded_pmc_traffic.json"))
print(t["workload"])
for k, v in t["kernels"].items(): print(k[:80], v["launches"], round(v["hbm_bytes"] / 1e6, 2), "MB")
PY
