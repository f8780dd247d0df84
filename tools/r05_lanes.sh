#!/bin/bash
# Vocoder lanes (pipeline.synthesize vocoder_lanes): bit-identity test, then the sharded config at 0 / 2 / 4 / 8 lanes
set -u
out=gpurun_out/r05lanes; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round5.py -q -m gpu -x -k "lanes or pipeline_exact" > $out/test.log 2>&1
tail -4 $out/test.log
for n in 0 4 2 8 0 4; do
  FDX_VOC_LANES=$n timeout 200 python bench.py --config sharded --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-extras > $out/sharded_l$n.json 2>> $out/err.log
  python - <<PY
import json
d = json.loads(open("$out/sharded_l$n.json").read().strip().splitlines()[-1])
print("lanes", $n, d["value"], d["ms_per_step"], d.get("stages_ms"))
PY
done
tail -5 $out/err.log
