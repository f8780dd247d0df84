#!/bin/bash
# Round-5 last call: the full -m gpu suite on the final commit (125 tests)
set -u
out=gpurun_out/r05fg2; mkdir -p $out
export TMPDIR=/tmp
( time timeout 620 python -m pytest tests -q -m gpu -x --durations=5 ) > $out/gpu_tests.log 2>&1
tail -12 $out/gpu_tests.log
