#!/bin/bash
# A/B of kernel tuning variants on the headline workload: un-profiled bench lines (no cpu baseline), interleaved twice
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
for v in "0 0" "1 0" "2 0" "3 0" "0 2" "3 2"; do
  set -- $v
  FDX_OUTP_VAR=$1 FDX_CONV_VAR=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie > $O/v_$1_$2_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("$O/v_$1_$2_$rep.json") if l.startswith("{")][-1])
print("outp_var $1 conv_var $2 rep $rep: ms/step", d["ms_per_step"], "conv us", d["roofline"]["avg_launch_us"], "outp us", d["other_kernels"][0]["avg_launch_us"], d["stages_ms"])
PY
done; done
