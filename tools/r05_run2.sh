#!/bin/bash
# round 5, GPU call 2: attention ubench (A/B, key-split sweep, cycle stamps), the ResBlock2 fixture inside the halo, tfdec end to end
set -u
export TMPDIR=/tmp
out=gpurun_out/r05b; mkdir -p $out
timeout 300 tools/ubench/attnqs 861 1 > $out/attnqs_T861_B1.txt 2>&1
timeout 300 tools/ubench/attnqs 861 8 > $out/attnqs_T861_B8.txt 2>&1
timeout 300 tools/ubench/attnqs 430 1 > $out/attnqs_T430_B1.txt 2>&1
cat $out/attnqs_T861_B1.txt; grep -v fp64 $out/attnqs_T861_B8.txt | head -5; head -12 $out/attnqs_T430_B1.txt
timeout 600 python -m pytest tests -x -q -m gpu -k "resblock2 or tfdec or cross" > $out/tests.log 2>&1
tail -3 $out/tests.log
timeout 200 python tools/tdbench.py 1 10 > $out/td_new.txt 2>&1; cat $out/td_new.txt
