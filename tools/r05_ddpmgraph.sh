#!/bin/bash
# EXPERIMENTAL build only (the recording was measured and not kept: profiles/r05_ddpm_graph_ab.txt); the timing legs still run on any build (FDX_NO_GRAPH is then a no-op for this sampler)
set -u
out=gpurun_out/r05ddpm; mkdir -p $out
export TMPDIR=/tmp
# (the experimental build also ran its own test here)

for sec in 2 5 10; do
  for ng in 0 1; do
    FDX_NO_GRAPH=$ng timeout 300 python bench.py --config ddpm1000 --batch 1 --seconds $sec --steps 3 --warmup 2 --no-cpu-baseline --no-prof --no-extras > $out/ddpm_b1_${sec}s_nograph$ng.json 2>> $out/err.log
    python - <<PY
import json
d = json.loads(open("$out/ddpm_b1_${sec}s_nograph$ng.json").read().strip().splitlines()[-1])
print("seconds", $sec, "FDX_NO_GRAPH", $ng, d["value"], d["ms_per_step"])
PY
  done
done
FDX_NO_GRAPH=0 timeout 400 python bench.py --config ddpm1000 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-extras > $out/ddpm_b16_graph.json 2>> $out/err.log
FDX_NO_GRAPH=1 timeout 400 python bench.py --config ddpm1000 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-extras > $out/ddpm_b16_eager.json 2>> $out/err.log
python - <<PY
import json
for n in ("graph", "eager"):
    d = json.loads(open("$out/ddpm_b16_%s.json" % n).read().strip().splitlines()[-1])
    print("batch 16", n, d["value"], d["ms_per_step"])
PY
grep -v amdgpu.ids $out/err.log | tail -5
