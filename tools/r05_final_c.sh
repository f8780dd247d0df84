#!/bin/bash
# Round-5 final evidence on the FINAL build: the full -m gpu suite, the default bench line, kernel stats of the two widening denoisers
set -u
out=gpurun_out/r05fc; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 ) > $out/gpu_tests.log 2>&1
tail -14 $out/gpu_tests.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 300 $out/bench_default.json; echo
for cfg in convnext tfdec; do
  args="--config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-extras"
  timeout 150 rocprofv3 --kernel-trace --stats -d $out/prof_$cfg -o kt -- python bench.py $args > $out/${cfg}_bench_under_rocprof.json 2> $out/${cfg}_rocprof.log
  python tools/prof_summary.py $out/prof_$cfg/kt_results.db > $out/${cfg}_kernel_stats.txt 2>&1
  rm -rf $out/prof_$cfg
  head -6 $out/${cfg}_kernel_stats.txt | cut -c1-170
done
