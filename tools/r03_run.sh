#!/bin/bash
# usage: tools/r03_run.sh <tag> <what...>   (GPU box; writes gpurun_out/<tag>/)
#   tests | tests-r3 | headline | sweep-outp | sweep-conv | f16s-small | and everything tools/r02_run.sh knows (prof-<cfg>, pmc-<cfg>, mfma, bench, rccl)
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r, o = d.get("roofline") or {}, (d.get("other_kernels") or [{}])[0]
    print(f"  value {d['value']}  ms/step {d['ms_per_step']}  stages {d.get('stages_ms')}  conv {r.get('avg_launch_us')} us ({r.get('frac')})  outp {o.get('avg_launch_us')} us ({o.get('frac')})")
    print("   ", (r.get("kernel") or "")[:150]); print("   ", (o.get("kernel") or "")[:150])
except Exception as e:
    print("  (no JSON line)", e)
PY
}
# do_prof <name> <bench.py args...>: rocprofv3 --kernel-trace --stats of the bench command -> $O/<name>_kernel_stats.txt (+ the bench line run under the profiler)
do_prof() { n=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$n -o p -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-pcie > $GRAFT_REPO_ROOT/$O/${n}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof_$n.log )
  db=$(ls $O/prof_$n/*/*.db $O/prof_$n/*.db 2>/dev/null | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $* --no-cpu-baseline --no-pcie   (round 3)"; python tools/prof_summary.py $db; } > $O/${n}_kernel_stats.txt; head -n 8 $O/${n}_kernel_stats.txt | cut -c1-200; rm -rf $O/prof_$n; }
# do_pmc <name> <workload tag for tools/pmc_traffic.py> <bench.py args...>: FETCH_SIZE and WRITE_SIZE in separate passes -> $O/<name>_pmc_traffic.json
do_pmc() { n=$1; wl=$2; shift 2
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && rocprofv3 --pmc $ctr -d $GRAFT_REPO_ROOT/$O/pmc_${n}_$ctr -o p -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-pcie --no-prof > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_${n}_$ctr.log )
  done
  f=$(ls $O/pmc_${n}_FETCH_SIZE/*/*.db $O/pmc_${n}_FETCH_SIZE/*.db 2>/dev/null | head -1); w=$(ls $O/pmc_${n}_WRITE_SIZE/*/*.db $O/pmc_${n}_WRITE_SIZE/*.db 2>/dev/null | head -1)
  python tools/pmc_traffic.py $f $w $wl > $O/${n}_pmc_traffic.json; head -c 400 $O/${n}_pmc_traffic.json; echo; rm -rf $O/pmc_${n}_FETCH_SIZE $O/pmc_${n}_WRITE_SIZE; }
for what in "$@"; do
  case $what in
    r3prof-headline) do_prof headline --config headline ;;
    r3prof-headline_fp16x3) do_prof headline_fp16x3 --config headline --storage fp16x3 ;;
    r3prof-ddpm1000) do_prof ddpm1000 --config ddpm1000 --interval 10 --steps 2 --warmup 1 ;;
    r3prof-ddpm1000_fp16x3) do_prof ddpm1000_fp16x3 --config ddpm1000 --storage fp16x3 --interval 10 --steps 2 --warmup 1 ;;
    r3prof-sharded) do_prof sharded --config sharded ;;
    r3prof-vocoder) do_prof vocoder --config vocoder ;;
    r3pmc-headline) do_pmc headline headline --config headline ;;
    r3pmc-headline_fp16x3) do_pmc headline_fp16x3 headline_fp16x3 --config headline --storage fp16x3 ;;
    r3pmc-ddpm1000) do_pmc ddpm1000 ddpm1000 --config ddpm1000 --interval 10 ;;
    r3bench) for c in headline vocoder sharded ddpm1000; do python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc $?"; line $O/bench_$c.json; done
             for c in headline sharded ddpm1000; do python bench.py --config $c --storage fp16x3 > $O/bench_${c}_fp16x3.json 2> $O/bench_${c}_fp16x3.err; echo "$c fp16x3 rc $?"; line $O/bench_${c}_fp16x3.json; done ;;
    defer) [ -n "$DEFER_TESTS" ] && FDX_DEFER_SKIP=20 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wavenet_forward or baseline_configs" 2>&1 | tail -n 3
           [ -n "$DEFER_TESTS" ] && FDX_DEFER_SKIP=5 FDX_DEFER_SIDE=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wavenet_forward or baseline_configs" 2>&1 | tail -n 3
           DS=${DEFER_SET:-0:27:0:0 20:27:0:0 20:24:0:0 20:17:0:0 20:24:17:0 20:24:24:0 5:24:0:0 5:24:0:1 5:24:17:1 4:27:17:1 10:24:17:1}
           for v in $DS; do
             G=$(echo $v | cut -d: -f1); SH=$(echo $v | cut -d: -f2); RS=$(echo $v | cut -d: -f3); SIDE=$(echo $v | cut -d: -f4); echo "FDX_DEFER_SKIP=$G SHAPE=$SH RES_SHAPE=$RS SIDE=$SIDE"
             FDX_DEFER_SKIP=$G FDX_DEFER_SHAPE=$SH FDX_DEFER_RES_SHAPE=$RS FDX_DEFER_SIDE=$SIDE python bench.py --no-cpu-baseline --no-pcie --steps 5 --warmup 2 > $O/defer_${G}_${SH}_${RS}_${SIDE}.json 2> $O/defer.err; line $O/defer_${G}_${SH}_${RS}_${SIDE}.json; done ;;
    tests-r3) python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s > $O/gpu_tests_r3.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests_r3.log; grep -v "^$" $O/gpu_tests_r3.log | tail -n 40 ;;
    tests) ( time python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -n 15 $O/gpu_tests.log ;;
    sweep-outp) for v in ${SWEEP_OUTP:-27 24 25 26 28 44 45}; do echo "FDX_OUTP_SHAPE=$v"; FDX_OUTP_SHAPE=$v python bench.py --no-cpu-baseline --no-pcie --steps 5 --warmup 2 > $O/sweep_outp_$v.json 2> $O/sweep_outp_$v.err; line $O/sweep_outp_$v.json; done ;;
    sweep-conv) for v in ${SWEEP_CONV:-27 24 25 26 28 44 45}; do echo "FDX_CONV_SHAPE=$v"; FDX_CONV_SHAPE=$v python bench.py --no-cpu-baseline --no-pcie --steps 5 --warmup 2 > $O/sweep_conv_$v.json 2> $O/sweep_conv_$v.err; line $O/sweep_conv_$v.json; done ;;
    sweep-voc) for v in ${SWEEP_VOC:-512 400 200 100}; do echo "FDX_NOSPLIT_MIN_WGS=$v"; FDX_NOSPLIT_MIN_WGS=$v python bench.py --no-cpu-baseline --no-pcie --no-prof --steps 5 --warmup 2 > $O/sweep_voc_$v.json 2> $O/sweep_voc_$v.err; line $O/sweep_voc_$v.json; done ;;
    ktrace-f16s) python tools/ktrace.py 1 20 fp16x3 > $O/ktrace_f16s64.txt 2>&1; head -n 8 $O/ktrace_f16s64.txt ;;
    ktrace) for v in 0 1; do FDX_LDS_OPS=$v python tools/ktrace.py 1 20 > $O/ktrace_fp32_lds$v.txt 2>&1; echo "FDX_LDS_OPS=$v"; head -n 8 $O/ktrace_fp32_lds$v.txt | cut -c1-200; done ;;
    cross) G="${CROSS_GEO:-1x108 1x215 1x430 1x645 1x861 2x430 2x861 3x861 4x861 6x861 8x861}"
           python tools/f16s_cross.py fp32 $G 2>&1 | grep CROSS | tee $O/cross_fp32.txt
           FDX_F16S_SMALL=2 FDX_BF16_LDS=1000000000 python tools/f16s_cross.py fp16x3 $G 2>&1 | grep CROSS | tee $O/cross_small.txt
           FDX_BF16_LDS=1 python tools/f16s_cross.py fp16x3 $G 2>&1 | grep CROSS | tee $O/cross_big.txt ;;
    nst) for c in 3 4; do for o in 3 4 6 8; do FDX_F16S_NST=$c FDX_F16S_NST_O=$o FDX_F16S_SMALL=2 FDX_BF16_LDS=1000000000 python tools/f16s_cross.py fp16x3 1x861 1x430 2>&1 | grep CROSS; done; done | tee $O/nst.txt ;;
    sweep-outp1) SWEEP_OUTP="14 15 16 17 18" bash tools/r03_run.sh $tag sweep-outp ;;
    tests-f16s-small) K=$(python -c "import tests.test_gpu_round2 as t; print(t.FP16X3_SUBSET)")
           ( time FDX_WAVENET_STORAGE=fp16x3 FDX_F16S_SMALL=2 FDX_BF16_LDS=1000000000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -q -x -s -k "$K" ) > $O/gpu_tests_f16s_small.log 2>&1
           echo "pytest rc $?" >> $O/gpu_tests_f16s_small.log; grep -E "passed|failed|error|rel err|chain|ragged" $O/gpu_tests_f16s_small.log | tail -n 40 ;;
    rect) for v in 0 2048; do echo "FDX_XCD_RECT=$v ddpm1000 (100 of 1000 steps), batch 16"; FDX_XCD_RECT=$v python bench.py --config ddpm1000 --steps 2 --warmup 1 --interval 10 --no-cpu-baseline > $O/rect_$v.json 2> $O/rect_$v.err; line $O/rect_$v.json; done
          for v in 0 1024; do echo "FDX_XCD_RECT=$v ddpm1000 (100 steps), batch 8"; FDX_XCD_RECT=$v python bench.py --config ddpm1000 --batch 8 --steps 2 --warmup 1 --interval 10 --no-cpu-baseline > $O/rect8_$v.json 2> $O/rect8_$v.err; line $O/rect8_$v.json; done
          for v in 0 512; do echo "FDX_XCD_RECT=$v headline batch 4"; FDX_XCD_RECT=$v python bench.py --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $O/rect4_$v.json 2> $O/rect4_$v.err; line $O/rect4_$v.json; done ;;
    rect-rows) for r in 2 4; do echo "FDX_XCD_RECT_ROWS=$r ddpm1000 (100 steps), batch 16"; FDX_XCD_RECT_ROWS=$r python bench.py --config ddpm1000 --steps 2 --warmup 1 --interval 10 --no-cpu-baseline > $O/rectrows_$r.json 2> $O/rectrows_$r.err; line $O/rectrows_$r.json
               FDX_XCD_RECT_ROWS=$r bash tools/r02_run.sh $tag pmc-ddpm1000 > /dev/null 2>&1; mv $O/ddpm1000_pmc_traffic.json $O/ddpm1000_pmc_traffic_rows$r.json
               python -c "
import json,sys
d=json.load(open('$O/ddpm1000_pmc_traffic_rows$r.json'))
for k,v in d['kernels'].items():
    if 'Gate' in k or 'ResSkip' in k: print('   ', k[:60], 'fetch MB', round(v['fetch_bytes']/1e6,1), 'write MB', round(v['write_bytes']/1e6,1), 'total', round(v['hbm_bytes']/1e6,1))
"; done ;;
    voc-seq) ( cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_voc -o p -- python $GRAFT_REPO_ROOT/tools/vocbench.py 1 > $GRAFT_REPO_ROOT/$O/vocbench.log 2>&1 )
             db=$(ls $O/prof_voc/*/*.db $O/prof_voc/*.db 2>/dev/null | head -1); python tools/prof_summary.py $db --sequence 150 > $O/voc_sequence.txt; tail -n 3 $O/vocbench.log; head -n 5 $O/voc_sequence.txt; rm -rf $O/prof_voc ;;
    fwd-seq) ( cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_fwd -o p -- python $GRAFT_REPO_ROOT/tools/fwdseq.py > $GRAFT_REPO_ROOT/$O/fwdseq.log 2>&1 )
             db=$(ls $O/prof_fwd/*/*.db $O/prof_fwd/*.db 2>/dev/null | head -1); python tools/prof_summary.py $db --sequence 50 > $O/fwd_sequence.txt; cut -c1-150 $O/fwd_sequence.txt | tail -n 52; rm -rf $O/prof_fwd ;;
    dynlds) for v in 0 60000 90000; do echo "FDX_DYN_LDS=$v"; FDX_DYN_LDS=$v python bench.py --no-cpu-baseline --no-pcie --steps 5 --warmup 2 > $O/dynlds_$v.json 2> $O/dynlds_$v.err; line $O/dynlds_$v.json | head -1; done ;;
    bigshapes) for v in 44 46 47 48; do echo "FDX_OUTP_SHAPE=$v batch 16"; FDX_OUTP_SHAPE=$v python bench.py --config ddpm1000 --steps 2 --warmup 1 --interval 10 --no-cpu-baseline > $O/big_outp_$v.json 2> $O/big_outp_$v.err; line $O/big_outp_$v.json | head -1; done
               for v in 46 47 48; do echo "FDX_CONV_SHAPE=$v batch 16"; FDX_CONV_SHAPE=$v python bench.py --config ddpm1000 --steps 2 --warmup 1 --interval 10 --no-cpu-baseline > $O/big_conv_$v.json 2> $O/big_conv_$v.err; line $O/big_conv_$v.json | head -1; done ;;
    clocks) for a in "1 20 fp32" "1 20 fp16x3" "16 20 fp32" "16 20 fp16x3" "16 20 bf16"; do echo "ktrace $a"; python tools/ktrace.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-220 | sed -n 1,8p | tee -a $O/ktrace_clocks.txt; done ;;
    ldsops) for v in 0 1; do echo "FDX_LDS_OPS=$v"; FDX_LDS_OPS=$v python bench.py --no-cpu-baseline --no-pcie --steps 5 --warmup 2 > $O/ldsops_$v.json 2> $O/ldsops_$v.err; line $O/ldsops_$v.json; done ;;
    test-shapes) python -m pytest tests/test_gpu_round2.py -m gpu -x -q -s -k every_conv_tile_shape 2>&1 | tail -n 5 ;;
    f16s-small) echo "--storage fp16x3"; python bench.py --storage fp16x3 --no-cpu-baseline --no-pcie > $O/bench_headline_fp16x3.json 2> $O/bench_headline_fp16x3.err; line $O/bench_headline_fp16x3.json ;;
    cross-nt) for v in 0 1; do FDX_F16S_NT=$v FDX_F16S_SMALL=2 FDX_BF16_LDS=1000000000 python tools/f16s_cross.py fp16x3 1x861 2x861 2>&1 | grep CROSS | sed "s/^/nt=$v /"; done | tee $O/cross_nt.txt ;;
    f16s-dbg) for v in 0 1 4 5 8; do echo "FDX_F16S_DBG=$v"; FDX_F16S_DBG=$v python bench.py --storage fp16x3 --no-cpu-baseline --no-pcie --steps 3 --warmup 1 > $O/f16s_dbg_$v.json 2> $O/f16s_dbg_$v.err; line $O/f16s_dbg_$v.json | head -1; done ;;
    cross-small) FDX_F16S_SMALL=2 FDX_BF16_LDS=1000000000 python tools/f16s_cross.py fp16x3 ${CROSS_GEO:-1x430 1x861 2x861} 2>&1 | grep CROSS | tee $O/cross_small2.txt ;;
    headline) python bench.py > $O/bench_headline.json 2> $O/bench_headline.err; echo "headline rc $?"; tail -n 3 $O/bench_headline.err; line $O/bench_headline.json ;;
    *) bash tools/r02_run.sh $tag $what ;;
  esac
done
