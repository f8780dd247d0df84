#!/bin/bash
# round 5, GPU call 8: clkwait with the combined-load variants (does the in-kernel clock reading follow board power?)
set -u
export TMPDIR=/tmp
out=gpurun_out/r05h; mkdir -p $out
dev=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
hw=$(ls -d $dev/hwmon/hwmon* 2>/dev/null | head -1)
( while true; do
    p=$(cat $hw/power1_average 2>/dev/null || cat $hw/power1_input 2>/dev/null || echo NA)
    sc=$(grep '\*' $dev/pp_dpm_sclk 2>/dev/null | tr -d '\n')
    echo "$(date +%s.%N) power_uW=$p sclk=$sc"
    sleep 0.01
  done ) > $out/power_log.txt 2>/dev/null &
logger=$!
timeout 120 tools/ubench/clkwait > $out/clkwait.txt 2>&1
kill $logger 2>/dev/null
python - <<'PY'
import re
out = "gpurun_out/r05h"
samples = []
for ln in open(f"{out}/power_log.txt"):
    m = re.match(r"([\d.]+) power_uW=(\S+) sclk=(.*)", ln)
    if m and m.group(2) not in ("NA", ""):
        try: samples.append((float(m.group(1)), float(m.group(2)) / 1e6, m.group(3)))
        except ValueError: pass
rows = []
for ln in open(f"{out}/clkwait.txt"):
    m = re.match(r"variant (\d) (.*?)\s+(long|chain)\s.*= ([\d.]+) GHz .*wall ([\d.]+) \.\. ([\d.]+)", ln)
    if m:
        t0, t1 = float(m.group(5)), float(m.group(6))
        s = [x for x in samples if t0 <= x[0] <= t1]
        pw = sum(x[1] for x in s) / len(s) if s else float("nan")
        rows.append(f"variant {m.group(1)} {m.group(2)[:56]:56s} {m.group(3):5s}: in-kernel {m.group(4)} GHz | {len(s):3d} sysfs samples, board power {pw:7.1f} W, sclk level {s[-1][2] if s else 'NA'}")
open(f"{out}/clkwait_power.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
grep -c . $out/clkwait.txt
