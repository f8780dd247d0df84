#!/bin/bash
set -u
out=gpurun_out/r05j; mkdir -p $out
timeout 200 tools/ubench/attnqs 861 1 > $out/attnqs.txt 2>&1
grep -v "^masked" $out/attnqs.txt | head -9 | cut -c1-260; grep -A8 "split 4 ways: mean" $out/attnqs.txt
timeout 200 tools/ubench/attnqs 861 8 > $out/attnqs_b8.txt 2>&1; grep "split 1 ways" $out/attnqs_b8.txt | head -1 | cut -c1-240
timeout 600 python -m pytest tests -x -q -m gpu -k "tfdec or cross or other_denoisers" > $out/tests.log 2>&1; tail -3 $out/tests.log
timeout 200 python tools/tdbench.py 1 10 2>&1 | tail -1
