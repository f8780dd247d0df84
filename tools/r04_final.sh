#!/bin/bash
# Round-4 final evidence in one GPU call: the full -m gpu suite (with the 128-column forced sweeps), the default bench line, and the headline's
# rocprofv3 kernel stats + PMC traffic -> gpurun_out/r04f/
set -u
out=gpurun_out/r04f; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1300 python -m pytest tests -q -m gpu -x --durations=12 ) > $out/gpu_tests.log 2>&1
tail -25 $out/gpu_tests.log
timeout 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 600 $out/bench_default.json; echo
tools/collect_profiles.sh headline r04f 3
