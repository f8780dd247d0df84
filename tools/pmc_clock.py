#!/usr/bin/env python3
"""Effective GPU clock of UN-instrumented kernels: GRBM_GUI_ACTIVE (graphics-engine busy cycles, sclk domain) per dispatch divided by that
dispatch's duration from the same rocprofv3 run.

    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d <dir> -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-extras --no-prof
    python tools/pmc_clock.py <dir>/pmc_results.db [counter]
"""
import sqlite3
import sys

db, counter = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "GRBM_GUI_ACTIVE")
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("PRAGMA table_info(counters_collection)")]
print("# counters_collection columns:", cols)
s, e = ("start", "end") if "start" in cols else (("start_timestamp", "end_timestamp") if "start_timestamp" in cols else (None, None))
if s is None:
    raise SystemExit("no timestamps in counters_collection: join with the kernel trace by dispatch id by hand")
rows = c.execute(f"select kernel_name, count(*), avg(value), avg({e} - {s}), min(value * 1.0 / ({e} - {s})), max(value * 1.0 / ({e} - {s})) "
                 f"from counters_collection where counter_name = ? and {e} > {s} group by kernel_name order by sum({e} - {s}) desc", (counter,)).fetchall()
print(f"{'kernel':<90s} {'calls':>6s} {counter + ' avg':>18s} {'avg ns':>10s} {'GHz avg':>8s} {'min':>6s} {'max':>6s}")
for k, n, v, ns, lo, hi in rows[:14]:
    print(f"{k.replace('void ', '').replace('fdx::', '')[:90]:<90s} {n:>6d} {v:>18.0f} {ns:>10.0f} {v / ns:>8.3f} {lo:>6.3f} {hi:>6.3f}")
