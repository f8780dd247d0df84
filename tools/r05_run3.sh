#!/bin/bash
# round 5, GPU call 3: tfdec with the hoisted cross-attention keys / values (parity subset + end-to-end time + kernel stats)
set -u
export TMPDIR=/tmp
out=gpurun_out/r05c; mkdir -p $out
timeout 600 python -m pytest tests -x -q -m gpu -k "tfdec or cross or transformer" > $out/tests.log 2>&1
tail -4 $out/tests.log
timeout 200 python tools/tdbench.py 1 10 > $out/td_new.txt 2>&1; cat $out/td_new.txt
timeout 200 python tools/tdbench.py 8 50 > $out/td_new_b8.txt 2>&1; cat $out/td_new_b8.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_td -o kt -- python $GRAFT_REPO_ROOT/tools/tdbench.py 1 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py /tmp/prof_td/kt_results.db > $out/tfdec_kernel_stats.txt 2>&1
head -12 $out/tfdec_kernel_stats.txt | cut -c1-200
