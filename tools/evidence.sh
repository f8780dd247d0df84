#!/bin/bash
# One parameterised evidence collector (round 6: replaces the one-shot tools/r0N_run*.sh scripts).  Runs ON the GPU box -- chain the
# sub-commands inside one `gpurun -- '...'` call; everything lands under gpurun_out/<tag>/, copy what is to be judged into profiles/.
#
#   tools/evidence.sh tests   <tag> [pytest args...]          pytest -m gpu with the given selection -> <tag>/tests.log
#   tools/evidence.sh bench   <tag> <config|default> [VAR=val ...] [-- bench args]   one bench.py line under the given environment -> <tag>/<config>[_VAR-val].json
#   tools/evidence.sh profile <tag> <config> [steps]          bench line + rocprofv3 kernel stats + the two PMC passes (tools/collect_profiles.sh)
#   tools/evidence.sh pmcmini <tag> <tfdec|sharded> [VAR=val ...]   FETCH_SIZE / WRITE_SIZE passes over tools/pmc_mini.py (configs whose full bench command kills rocprofv3)
#   tools/evidence.sh pmc     <tag> <name> "<counters>" [VAR=val ...] -- <command...>   one rocprofv3 --pmc pass (kernel trace only) over any command -> <tag>/<name>_pmc.txt
#   tools/evidence.sh stats   <tag> <name> [VAR=val ...] -- <command...>   rocprofv3 --kernel-trace --stats of any command -> <tag>/<name>_kernel_stats.txt (+ _sequence.txt)
#   tools/evidence.sh run     <tag> <name> [VAR=val ...] -- <command...>   any command, stdout+stderr -> <tag>/<name>.txt
set -u
export TMPDIR=/tmp
sub=$1; tag=$2; shift 2
out=gpurun_out/$tag; mkdir -p "$out"
envs=(); 
take_envs() { envs=(); rest=(); local seen=0; for a in "$@"; do if [ $seen = 0 ] && [[ "$a" == *=* ]] && [[ "$a" != -* ]]; then envs+=("$a"); elif [ "$a" = "--" ] && [ $seen = 0 ]; then seen=1; else seen=1; rest+=("$a"); fi; done; }
suffix() { local s=""; for e in "${envs[@]:-}"; do [ -n "$e" ] && [[ "$e" != FDX_LIB_PATH=* ]] && s="${s}_${e//=/-}"; done; s="${s//\//_}"; echo "${s:0:80}"; }
case $sub in
  tests)
    where=tests; for a in "$@"; do [[ "$a" == tests/* ]] && where=""; done      # a file / node id among the arguments replaces the whole directory
    timeout ${FDX_TEST_TIMEOUT:-1500} python -m pytest $where -m gpu -q "$@" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log; tail -4 $out/tests.log ;;
  bench)
    cfg=$1; shift; take_envs "$@"
    args="--steps ${FDX_STEPS:-3} --warmup 1 --no-cpu-baseline --no-pcie --no-extras"; [ "$cfg" != default ] && args="--config $cfg $args"
    [ "$cfg" = default ] && args=""
    f=$out/${cfg}$(suffix).json
    env "${envs[@]:-FDX_NOP=1}" timeout ${FDX_BENCH_TIMEOUT:-600} python bench.py $args ${rest[@]+"${rest[@]}"} > $f 2> ${f%.json}.err
    python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1], "value", d.get("value"), "ms", d.get("ms_per_step"), "kernel", r.get("avg_launch_us"), "us frac", r.get("frac"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
    ;;
  profile)
    tools/collect_profiles.sh "$1" "$tag" "${2:-3}" ;;
  pmcmini)
    cfg=$1; shift; take_envs "$@"
    for c in FETCH_SIZE WRITE_SIZE; do
      env "${envs[@]:-FDX_NOP=1}" timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${tag}_$c -o pmc -- python tools/pmc_mini.py $cfg > $out/${cfg}$(suffix)_pmc_$c.log 2>&1
    done
    f=$out/${cfg}$(suffix)_pmc_traffic.json
    python tools/pmc_traffic.py /tmp/pmc_${tag}_FETCH_SIZE/pmc_results.db /tmp/pmc_${tag}_WRITE_SIZE/pmc_results.db $cfg > $f 2> ${f%.json}.err
    rm -rf /tmp/pmc_${tag}_FETCH_SIZE /tmp/pmc_${tag}_WRITE_SIZE
    python - "$f" <<'PY'
import json, sys
t = json.load(open(sys.argv[1]))
for k, v in t["kernels"].items():
    print(k[:90], v["launches"], round(v["hbm_bytes"] / 1e6, 2), "MB")
PY
    ;;
  pmc)
    name=$1; counters=$2; shift 2; take_envs "$@"
    env "${envs[@]:-FDX_NOP=1}" timeout ${FDX_PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $counters -d /tmp/pmc_${tag}_$name -o pmc -- "${rest[@]}" > $out/${name}_pmc.log 2>&1
    python tools/prof_summary.py /tmp/pmc_${tag}_$name/pmc_results.db --pmc > $out/${name}_pmc.txt 2>&1
    rm -rf /tmp/pmc_${tag}_$name
    grep -E "k_attn_qs|EpiResLN|4, EpiBias|EpiGate|EpiResSkip|EpiScaleRes" $out/${name}_pmc.txt | cut -c1-150 | head -40 ;;
  stats)
    name=$1; shift; take_envs "$@"
    env "${envs[@]:-FDX_NOP=1}" timeout ${FDX_PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_$name -o kt -- "${rest[@]}" > $out/${name}.log 2>&1
    python tools/prof_summary.py /tmp/prof_${tag}_$name/kt_results.db > $out/${name}_kernel_stats.txt 2>&1
    python tools/prof_summary.py /tmp/prof_${tag}_$name/kt_results.db --sequence ${FDX_SEQ:-300} > $out/${name}_sequence.txt 2>&1
    rm -rf /tmp/prof_${tag}_$name
    head -10 $out/${name}_kernel_stats.txt | cut -c1-175 ;;
  run)
    name=$1; shift; take_envs "$@"
    env "${envs[@]:-FDX_NOP=1}" timeout ${FDX_RUN_TIMEOUT:-900} "${rest[@]}" > $out/${name}$(suffix).txt 2>&1; echo "rc=$?" >> $out/${name}$(suffix).txt
    grep -v amdgpu.ids $out/${name}$(suffix).txt | tail -${FDX_TAIL:-6} ;;
  *) echo "unknown sub-command $sub"; exit 2 ;;
esac
