#!/bin/bash
# Round-5 last call: the default bench line on the committed final build (attention A/B arm removed, PMC traffic files for every row in place)
set -u
out=gpurun_out/r05fd; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 200 $out/bench_default.json; echo
