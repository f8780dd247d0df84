#!/bin/bash
# usage: tools/r03_run.sh <tag> <what...>   (GPU box; writes gpurun_out/<tag>/)
#   tests | tests-r3 | headline | sweep-outp | sweep-conv | f16s-small | and everything tools/r02_run.sh knows (prof-<cfg>, pmc-<cfg>, mfma, bench, rccl)
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r, o = d.get("roofline") or {}, (d.get("other_kernels") or [{}])[0]
    print(f"  value {d['value']}  ms/step {d['ms_per_step']}  stages {d.get('stages_ms')}  conv {r.get('avg_launch_us')} us ({r.get('frac')})  outp {o.get('avg_launch_us')} us ({o.get('frac')})")
    print("   ", (r.get("kernel") or "")[:150]); print("   ", (o.get("kernel") or "")[:150])
except Exception as e:
    print("  (no JSON line)", e)
PY
}
for what in "$@"; do
  case $what in
    tests-r3) python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s > $O/gpu_tests_r3.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests_r3.log; grep -v "^$" $O/gpu_tests_r3.log | tail -n 40 ;;
    tests) ( time python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -n 15 $O/gpu_tests.log ;;
    sweep-outp) for v in ${SWEEP_OUTP:-27 24 25 26 28 44 45}; do echo "FDX_OUTP_SHAPE=$v"; FDX_OUTP_SHAPE=$v python bench.py --no-cpu-baseline --no-pcie --steps 5 --warmup 2 > $O/sweep_outp_$v.json 2> $O/sweep_outp_$v.err; line $O/sweep_outp_$v.json; done ;;
    sweep-conv) for v in ${SWEEP_CONV:-27 24 25 26 28 44 45}; do echo "FDX_CONV_SHAPE=$v"; FDX_CONV_SHAPE=$v python bench.py --no-cpu-baseline --no-pcie --steps 5 --warmup 2 > $O/sweep_conv_$v.json 2> $O/sweep_conv_$v.err; line $O/sweep_conv_$v.json; done ;;
    sweep-voc) for v in ${SWEEP_VOC:-512 400 200 100}; do echo "FDX_NOSPLIT_MIN_WGS=$v"; FDX_NOSPLIT_MIN_WGS=$v python bench.py --no-cpu-baseline --no-pcie --no-prof --steps 5 --warmup 2 > $O/sweep_voc_$v.json 2> $O/sweep_voc_$v.err; line $O/sweep_voc_$v.json; done ;;
    ktrace-f16s) FDX_F16S_SMALL=1 python tools/ktrace.py 1 20 fp16x3 > $O/ktrace_f16s64.txt 2>&1; head -n 8 $O/ktrace_f16s64.txt ;;
    ktrace) python tools/ktrace.py 1 20 > $O/ktrace_fp32.txt 2>&1; head -n 8 $O/ktrace_fp32.txt ;;
    f16s-small) for v in 1; do echo "FDX_F16S_SMALL=$v --storage fp16x3"; FDX_F16S_SMALL=$v python bench.py --storage fp16x3 --no-cpu-baseline --no-pcie > $O/bench_headline_fp16x3.json 2> $O/bench_headline_fp16x3.err; line $O/bench_headline_fp16x3.json; done ;;
    headline) python bench.py > $O/bench_headline.json 2> $O/bench_headline.err; echo "headline rc $?"; tail -n 3 $O/bench_headline.err; line $O/bench_headline.json ;;
    *) bash tools/r02_run.sh $tag $what ;;
  esac
done
