#!/bin/bash
# Round-5 final evidence, call B: rocprofv3 kernel stats + PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the headline, vocoder,
# sharded, convnext and tfdec bench configs -> gpurun_out/r05fb/ (every leg under its own timeout)
set -u
export TMPDIR=/tmp FDX_PROF_TIMEOUT=170
for cfg in headline tfdec convnext sharded vocoder; do
  tools/collect_profiles.sh $cfg r05fb 3
done
ls gpurun_out/r05fb
