#!/bin/bash
# Round-5 final evidence, call A: the full -m gpu suite and the default bench line -> gpurun_out/r05fa/
set -u
out=gpurun_out/r05fa; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 ) > $out/gpu_tests.log 2>&1
tail -22 $out/gpu_tests.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 400 $out/bench_default.json; echo
