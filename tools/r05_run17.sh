#!/bin/bash
set -u
out=gpurun_out/r05q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > $out/tests.log 2>&1; tail -12 $out/tests.log
timeout 600 python -m pytest tests -x -q -m gpu -k "ragged or tfdec or cross" --deselect tests/test_gpu_round5.py > $out/tests_reg.log 2>&1; tail -3 $out/tests_reg.log
