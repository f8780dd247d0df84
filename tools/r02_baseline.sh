#!/bin/bash
# round-2 first GPU call: sanity + baselines of everything the round intends to move
set -x
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
FDX_FORCE_PROCESS_GROUP=1 python bench.py --steps 10 --warmup 3 > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err; echo "rc $?" >> $O/bench_rccl_world1.err
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python tools/ktrace.py 1 > $O/ktrace_b1.txt 2>&1
python tools/c4bench.py > $O/c4.txt 2>&1
python tools/vocbench.py 32 --hop256 > $O/voc32_256.txt 2>&1
python tools/vocbench.py 32 > $O/voc32_512.txt 2>&1
python tools/c5bench.py 16 100 > $O/c5_100.txt 2>&1
python tools/c5bench.py 16 100 bf16 > $O/c5_100_bf16.txt 2>&1
tail -3 $O/*.txt $O/gpu_tests.log
cat $O/bench_rccl_world1.err | tail -20
