#!/bin/bash
set -u
out=gpurun_out/r05s; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -s -k "30_s or layouts" > $out/tests.log 2>&1; grep "tfdec dim\|passed\|failed\|Error" $out/tests.log | head
