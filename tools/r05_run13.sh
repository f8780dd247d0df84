#!/bin/bash
set -u
out=gpurun_out/r05m; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu -k "tfdec or cross or other_denoisers or pipeline or convnext" > $out/tests.log 2>&1
tail -3 $out/tests.log
timeout 100 tools/ubench/attnqs 861 1 > $out/attnqs.txt 2>&1; head -6 $out/attnqs.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
