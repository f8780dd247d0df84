#!/bin/bash
# Refresh of the round-4 evidence after the late kernel changes (attention, GELU, EpiResblock prefetch): the default bench line and the two
# widening rows' rocprofv3 kernel stats -> gpurun_out/r04r/
set -u
out=gpurun_out/r04r; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
for cfg in tfdec convnext; do
  args="--config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-extras"
  timeout 150 rocprofv3 --kernel-trace --stats -d $out/prof_$cfg -o kt -- python bench.py $args > $out/${cfg}_bench_under_rocprof.json 2> $out/${cfg}_rocprof.log
  python tools/prof_summary.py $out/prof_$cfg/kt_results.db > $out/${cfg}_kernel_stats.txt 2>&1
  rm -rf $out/prof_$cfg
  head -6 $out/${cfg}_kernel_stats.txt | cut -c1-170
done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04r/bench_default.json"))
print(d["value"], d["ms_per_step"], d["stages_ms"])
for sec in ("configs","widening"):
    for k,v in d[sec].items(): print(k, v["value"], v["ms_per_step"], v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("avg_launch_us"))
PY
