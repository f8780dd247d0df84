#!/bin/bash
# Round-5 closing call after the vocoder lanes (host-side change only): the serving-loop tests, smoke, and the default bench line
set -u
out=gpurun_out/r05ff; mkdir -p $out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x -k "pipeline or lanes or ragged or serving or isolate or svc_inference" ) > $out/gpu_tests_serving.log 2>&1
tail -6 $out/gpu_tests_serving.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python - <<PY
import json
t = open("$out/bench_default.json").read().strip().splitlines()[-1]
d = json.loads(t)
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: v.get("value") for k, v in d["configs"].items()})
PY
