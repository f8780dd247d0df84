#!/bin/bash
# round 5, GPU call 4: 128-row workgroup tiles (MT = 2) + the 2-D tile -> XCD map of the split-K family: parity, bit-identity, A/B times
set -u
export TMPDIR=/tmp
out=gpurun_out/r05d; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu -k "two_mtile or tfdec or cross or convnext" > $out/tests.log 2>&1
tail -4 $out/tests.log
for v in "default:" "mt2off:FDX_MT2_MAX_WGS=0" "rectoff:FDX_SPLITK_RECT=0" "bothoff:FDX_MT2_MAX_WGS=0 FDX_SPLITK_RECT=0"; do
  tag=${v%%:*}; envs=${v#*:}
  env $envs timeout 200 python tools/cnbench.py 1 10 > $out/cn_$tag.txt 2>&1
  env $envs timeout 200 python tools/tdbench.py 1 10 > $out/td_$tag.txt 2>&1
  echo "$tag: $(tail -1 $out/cn_$tag.txt)"; echo "$tag: $(tail -1 $out/td_$tag.txt)"
done
for v in "default:" "rectoff:FDX_SPLITK_RECT=0"; do
  tag=${v%%:*}; envs=${v#*:}
  env $envs timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie --no-extras > $out/headline_$tag.json 2> $out/headline_$tag.err
  python - <<PY
import json
d = json.loads(open("$out/headline_$tag.json").read().strip().splitlines()[-1])
print("$tag headline", d["value"], d["ms_per_step"])
PY
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cn -o kt -- python $GRAFT_REPO_ROOT/tools/cnbench.py 1 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py /tmp/prof_cn/kt_results.db > $out/convnext_kernel_stats.txt 2>&1
head -8 $out/convnext_kernel_stats.txt | cut -c1-200
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_td -o kt -- python $GRAFT_REPO_ROOT/tools/tdbench.py 1 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py /tmp/prof_td/kt_results.db > $out/tfdec_kernel_stats.txt 2>&1
head -9 $out/tfdec_kernel_stats.txt | cut -c1-200
