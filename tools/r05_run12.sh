#!/bin/bash
# round 5, GPU call 12: prologue clean-ups of the small latency-bound kernels (dwconv + stats, LayerNorm, attention combine): parity subset + times
set -u
export TMPDIR=/tmp
out=gpurun_out/r05l; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu -k "tfdec or cross or convnext or other_denoisers or pipeline_exact" > $out/tests.log 2>&1
tail -3 $out/tests.log
timeout 200 python tools/cnbench.py 1 10 2>&1 | tail -1
timeout 200 python tools/tdbench.py 1 10 2>&1 | tail -1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cn -o kt -- python $GRAFT_REPO_ROOT/tools/cnbench.py 1 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py /tmp/prof_cn/kt_results.db > $out/convnext_kernel_stats.txt 2>&1
head -5 $out/convnext_kernel_stats.txt | cut -c1-180
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_td -o kt -- python $GRAFT_REPO_ROOT/tools/tdbench.py 1 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py /tmp/prof_td/kt_results.db > $out/tfdec_kernel_stats.txt 2>&1
head -8 $out/tfdec_kernel_stats.txt | cut -c1-180
