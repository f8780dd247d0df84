#!/bin/bash
set -u
out=gpurun_out/r05i; mkdir -p $out
timeout 200 tools/ubench/attnqs_fine 861 1 > $out/attnqs_fine.txt 2>&1
grep -A9 "keys split 4 ways: mean" $out/attnqs_fine.txt; grep "split 4 ways (" $out/attnqs_fine.txt | head -1 | cut -c1-200
