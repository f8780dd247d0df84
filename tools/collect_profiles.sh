#!/bin/bash
# One GPU call's worth of round-N evidence for a bench config: the bench line, rocprofv3 kernel stats of the same command, and the two PMC
# passes (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only) -> gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
#   tools/collect_profiles.sh <config> <tag> [steps] [extra bench args]
set -u
# every leg runs under its own timeout: a PMC pass over the tfdec graph once sat on 8364 incomplete dispatches for 24 GPU-minutes (r04)
cfg=$1; tag=$2; steps=${3:-3}; shift; shift; shift || true
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
args="--config $cfg --steps $steps --warmup 1 --no-cpu-baseline --no-pcie --no-extras $*"
timeout ${FDX_PROF_TIMEOUT:-150} python bench.py $args > $out/${cfg}_bench.json 2> $out/${cfg}_bench.err
timeout ${FDX_PROF_TIMEOUT:-150} rocprofv3 --kernel-trace --stats -d $out/prof_$cfg -o kt -- python bench.py $args > $out/${cfg}_bench_under_rocprof.json 2> $out/${cfg}_rocprof.log
python tools/prof_summary.py $out/prof_$cfg/kt_results.db > $out/${cfg}_kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout ${FDX_PROF_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${cfg}_$c -o pmc -- python bench.py $args --no-prof > /dev/null 2> $out/${cfg}_pmc_$c.log
done
python tools/pmc_traffic.py $out/pmc_${cfg}_FETCH_SIZE/pmc_results.db $out/pmc_${cfg}_WRITE_SIZE/pmc_results.db $cfg > $out/${cfg}_pmc_traffic.json 2> $out/${cfg}_pmc_traffic.err
rm -rf $out/prof_$cfg $out/pmc_${cfg}_FETCH_SIZE $out/pmc_${cfg}_WRITE_SIZE
head -8 $out/${cfg}_kernel_stats.txt | cut -c1-170
