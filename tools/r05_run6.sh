#!/bin/bash
# round 5, GPU call 6 (two time-boxed experiments): (a) does s_memtime advance in wait states / what clock do mixed kernels see,
# with board power + sclk sampled from sysfs; (b) LDS bank-conflict counters of the fp32 LDS-DMA GEMM cuts (tools/ubench/f32lds).
set -u
export TMPDIR=/tmp
out=gpurun_out/r05f; mkdir -p $out
# ---- (a)
dev=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
hw=$(ls -d $dev/hwmon/hwmon* 2>/dev/null | head -1)
echo "device $dev hwmon $hw: $(ls $hw 2>/dev/null | tr '\n' ' ')" > $out/power_log.txt
( while true; do
    p=$(cat $hw/power1_average 2>/dev/null || cat $hw/power1_input 2>/dev/null || echo NA)
    f=$(cat $hw/freq1_input 2>/dev/null || echo NA)
    sc=$(grep '\*' $dev/pp_dpm_sclk 2>/dev/null | tr -d '\n')
    echo "$(date +%s.%N) power_uW=$p freq1_Hz=$f sclk=$sc"
    sleep 0.02
  done ) >> $out/power_log.txt 2>/dev/null &
logger=$!
timeout 120 tools/ubench/clkwait > $out/clkwait.txt 2>&1
kill $logger 2>/dev/null
cat $out/clkwait.txt
python - <<'PY'
import re
out = "gpurun_out/r05f"
samples = []
for ln in open(f"{out}/power_log.txt"):
    m = re.match(r"([\d.]+) power_uW=(\S+) freq1_Hz=(\S+) sclk=(.*)", ln)
    if m and m.group(2) not in ("NA", ""):
        try:
            samples.append((float(m.group(1)), float(m.group(2)) / 1e6, m.group(3), m.group(4)))
        except ValueError:
            pass
rows = []
for ln in open(f"{out}/clkwait.txt"):
    m = re.match(r"variant (\d) (.*?)\s+(long|chain)\s.*wall ([\d.]+) \.\. ([\d.]+)", ln)
    if m:
        t0, t1 = float(m.group(4)), float(m.group(5))
        s = [x for x in samples if t0 <= x[0] <= t1]
        pw = sum(x[1] for x in s) / len(s) if s else float("nan")
        fr = [float(x[2]) / 1e6 for x in s if x[2] not in ("NA", "")]
        rows.append(f"variant {m.group(1)} {m.group(3):5s}: {len(s):3d} sysfs samples in the window, mean board power {pw:7.1f} W, mean freq1_input {sum(fr)/len(fr) if fr else float('nan'):7.1f} MHz, sclk level {s[-1][3] if s else 'NA'}")
open(f"{out}/clkwait_power.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
# ---- (b)
for mode in 0 8; do
  cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc_f32lds_$mode -o pmc -- $GRAFT_REPO_ROOT/tools/ubench/f32lds $mode > $GRAFT_REPO_ROOT/$out/f32lds_run_$mode.txt 2>&1
  cd $GRAFT_REPO_ROOT
  for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python tools/pmc_clock.py /tmp/pmc_f32lds_$mode/pmc_results.db $c > $out/f32lds_${mode}_$c.txt 2>&1; done
  tail -4 $out/f32lds_${mode}_SQ_LDS_BANK_CONFLICT.txt | cut -c1-200; tail -4 $out/f32lds_${mode}_SQ_LDS_IDX_ACTIVE.txt | cut -c1-200
done
# the product's attention kernel on the same two counters (its K / V tiles are staged through LDS as well)
cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc_attn -o pmc -- $GRAFT_REPO_ROOT/tools/ubench/attnqs 861 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python tools/pmc_clock.py /tmp/pmc_attn/pmc_results.db $c > $out/attnqs_$c.txt 2>&1; tail -5 $out/attnqs_$c.txt | cut -c1-200; done
