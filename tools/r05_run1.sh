#!/bin/bash
# round 5, GPU call 1: the query-split attention kernel (parity subset + A/B against round 4's kernel + key-split sweep + kernel stats)
set -u
export TMPDIR=/tmp
out=gpurun_out/r05a; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu -k "resblock2 or tfdec or cross or transformer" > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log
tail -5 $out/tests.log
FDX_ATTN=old timeout 200 python tools/tdbench.py 1 10 > $out/td_old.txt 2>&1
timeout 200 python tools/tdbench.py 1 10 > $out/td_new.txt 2>&1
for ks in 2 3 5 6 8; do FDX_ATTN_KSPLIT=$ks timeout 200 python tools/tdbench.py 1 10 > $out/td_new_ks$ks.txt 2>&1; done
timeout 200 python tools/tdbench.py 8 50 > $out/td_new_b8.txt 2>&1
FDX_ATTN=old timeout 200 python tools/tdbench.py 8 50 > $out/td_old_b8.txt 2>&1
cat $out/td_*.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_td -o kt -- python $GRAFT_REPO_ROOT/tools/tdbench.py 1 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py /tmp/prof_td/kt_results.db > $out/tfdec_kernel_stats.txt 2>&1
head -12 $out/tfdec_kernel_stats.txt | cut -c1-200
