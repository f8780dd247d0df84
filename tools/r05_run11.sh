#!/bin/bash
set -u
out=gpurun_out/r05k; mkdir -p $out
timeout 200 tools/ubench/attnqs 861 1 > $out/attnqs.txt 2>&1
grep "split 4 ways (\|k_attn<64,1>" $out/attnqs.txt | head -3 | cut -c1-230; grep -A7 "split 4 ways: mean" $out/attnqs.txt
timeout 200 tools/ubench/attnqs_fine 861 1 > $out/attnqs_fine.txt 2>&1; grep -A8 "split 4 ways: mean" $out/attnqs_fine.txt
timeout 200 tools/ubench/attnqs 861 8 > $out/attnqs_b8.txt 2>&1; grep "split 1 ways" $out/attnqs_b8.txt | head -1 | cut -c1-240
