#!/bin/bash
# usage: tools/r02_run.sh <tag> <what...>   -- what: tests bench rccl prof-<cfg> pmc-<cfg>
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    tests) python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -n 15 $O/gpu_tests.log ;;
    tests-new) python -m pytest tests/test_gpu_round2.py -m gpu -x -q -s > $O/gpu_tests_new.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests_new.log; tail -n 30 $O/gpu_tests_new.log ;;
    bench) for c in headline vocoder sharded ddpm1000; do python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc $?"; tail -n 3 $O/bench_$c.err; done
           python bench.py --config ddpm1000 --storage bf16 > $O/bench_ddpm1000_bf16.json 2> $O/bench_ddpm1000_bf16.err; echo "bf16 rc $?" ;;
    headline) python bench.py > $O/bench_headline.json 2> $O/bench_headline.err; echo "headline rc $?"; tail -n 3 $O/bench_headline.err ;;
    rccl) FDX_FORCE_PROCESS_GROUP=1 NCCL_DEBUG=INFO python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rccl_world1.json 2> $O/rccl_world1.err; echo "rccl rc $?" ;;
    prof-*) c=${what#prof-}; extra=""; [ "$c" = "ddpm1000_bf16" ] && { c=ddpm1000; extra="--storage bf16"; }; [ "$c" = "ddpm1000_fp16x3" ] && { c=ddpm1000; extra="--storage fp16x3"; }
            ( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_${what#prof-} -o p -- python $GRAFT_REPO_ROOT/bench.py --config $c $extra --no-cpu-baseline --no-pcie > $GRAFT_REPO_ROOT/$O/bench_${what#prof-}_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof_${what#prof-}.log )
            db=$(ls $O/prof_${what#prof-}/*/*.db $O/prof_${what#prof-}/*.db 2>/dev/null | head -1); python tools/prof_summary.py $db > $O/${what#prof-}_kernel_stats.txt; head -n 12 $O/${what#prof-}_kernel_stats.txt; rm -rf $O/prof_${what#prof-} ;;
    pmc-*) c=${what#pmc-}; cfg=$c; extra=""; [ "$c" = "ddpm1000_bf16" ] && { cfg=ddpm1000; extra="--storage bf16"; }; [ "$c" = "ddpm1000_fp16x3" ] && { cfg=ddpm1000; extra="--storage fp16x3"; }
           steps="--steps 1 --warmup 1"; [ "$cfg" = "ddpm1000" ] && steps="--steps 1 --warmup 1 --interval 10"   # 100 of the 1000 steps: same launches
           for ctr in FETCH_SIZE WRITE_SIZE; do
             ( cd /tmp && rocprofv3 --pmc $ctr -d $GRAFT_REPO_ROOT/$O/pmc_${c}_$ctr -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg $extra $steps --no-cpu-baseline --no-pcie --no-prof > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_${c}_$ctr.log )
           done
           f=$(ls $O/pmc_${c}_FETCH_SIZE/*/*.db $O/pmc_${c}_FETCH_SIZE/*.db 2>/dev/null | head -1); w=$(ls $O/pmc_${c}_WRITE_SIZE/*/*.db $O/pmc_${c}_WRITE_SIZE/*.db 2>/dev/null | head -1)
           python tools/pmc_traffic.py $f $w $c > $O/${c}_pmc_traffic.json; head -c 600 $O/${c}_pmc_traffic.json; rm -rf $O/pmc_${c}_FETCH_SIZE $O/pmc_${c}_WRITE_SIZE ;;
    mfma) ( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie --no-prof > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_mfma.log )
          db=$(ls $O/pmc_mfma/*/*.db $O/pmc_mfma/*.db 2>/dev/null | head -1); python tools/prof_summary.py $db --pmc > $O/pmc_mfma.txt; head -n 30 $O/pmc_mfma.txt; rm -rf $O/pmc_mfma ;;
    *) echo "unknown $what" ;;
  esac
done
