#!/bin/bash
# Round-5 closing call on the committed final build: full -m gpu suite, default bench line, the two opt-in storage modes on the headline geometry
set -u
out=gpurun_out/r05fe; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=5 ) > $out/gpu_tests.log 2>&1
tail -9 $out/gpu_tests.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 200 $out/bench_default.json; echo
for st in fp16x3 bf16; do
  timeout 300 python bench.py --storage $st --no-cpu-baseline --no-pcie --no-extras > $out/bench_headline_$st.json 2> $out/bench_headline_$st.err
  python - <<PY
import json
d = json.loads(open("$out/bench_headline_$st.json").read().strip().splitlines()[-1])
print("$st", d["value"], d["ms_per_step"], d["dtype"])
PY
done
