#!/bin/bash
# out-projection family / shape A/B: headline (batch 1) and the ddpm shape (batch 16, 100 steps)
for rep in 1 2; do
for v in 0 auto 44 27; do
  if [ $v = auto ]; then unset FDX_OUTP_SHAPE; else export FDX_OUTP_SHAPE=$v; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('headline outp $v rep $rep:', d['ms_per_step'], 'conv us', d['roofline']['avg_launch_us'], 'outp us', d['other_kernels'][0]['avg_launch_us'], d['stages_ms'])"
done
for v in 0 auto 47; do
  if [ $v = auto ]; then unset FDX_OUTP_SHAPE; else export FDX_OUTP_SHAPE=$v; fi
  python bench.py --config ddpm1000 --interval 10 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B16 outp $v rep $rep:', d['ms_per_step'], 'conv us', d['roofline']['avg_launch_us'], 'outp us', d['other_kernels'][0]['avg_launch_us'])"
done
done
