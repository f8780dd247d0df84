#!/bin/bash
# round 5, GPU call 5: exact-ragged batches for ConvNext / transformer + regression subset
set -u
export TMPDIR=/tmp
out=gpurun_out/r05e; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > $out/tests_r5.log 2>&1
tail -15 $out/tests_r5.log
timeout 900 python -m pytest tests -x -q -m gpu -k "ragged or tfdec or cross or convnext" --deselect tests/test_gpu_round5.py > $out/tests_reg.log 2>&1
tail -4 $out/tests_reg.log
