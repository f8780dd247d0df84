#!/bin/bash
# A/B of the bf16 storage mode's two kernels (register-direct vs LDS-tiled) on 100 of the 1000 DDPM steps + the ktrace breakdown
for l in ${1:-0 256}; do
  FDX_BF16_LDS=$l python bench.py --config ddpm1000 --storage bf16 --interval 10 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('lds=$l ms/100steps', d['ms_per_step'], 'conv us', r['avg_launch_us'], 'frac', r['frac'], 'outp us', [k.get('avg_launch_us') for k in (d.get('other_kernels') or [])])"
done
python tools/ktrace.py 16 20 bf16 2>&1 | tail -5
