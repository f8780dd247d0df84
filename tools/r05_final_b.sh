#!/bin/bash
# Round-5 final evidence, call B: the default bench line (widening rows profiled outside their timed region), then rocprofv3 kernel stats + PMC
# traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the headline, convnext, sharded, vocoder and tfdec bench configs -> gpurun_out/r05fb/
set -u
export TMPDIR=/tmp FDX_PROF_TIMEOUT=120
out=gpurun_out/r05fb; mkdir -p $out
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 300 $out/bench_default.json; echo
for cfg in headline convnext sharded vocoder tfdec; do
  tools/collect_profiles.sh $cfg r05fb 3
done
ls $out
