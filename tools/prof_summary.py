#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats` on ROCm 7.2)
as a per-kernel table: calls, total / average / min / max duration (us), share of GPU time.

    python tools/prof_summary.py gpurun_out/prof/x_results.db [> profiles/rNN_kernel_stats.txt]
With --pmc the per-kernel mean of every collected counter is printed instead."""
import sqlite3
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    db = sqlite3.connect(sys.argv[1])
    if "--pmc" in sys.argv:
        rows = db.execute("""select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection
                             group by kernel_name, counter_name order by kernel_name, counter_name""").fetchall() \
            if has(db, "counters_collection", "kernel_name") else pmc_fallback(db)
        print(f"{'kernel':<70} {'counter':<28} {'n':>7} {'mean':>16} {'sum':>18}")
        for name, ctr, n, mean, tot in rows:
            print(f"{short(name):<70} {ctr:<28} {n:>7} {mean:>16.1f} {tot:>18.0f}")
        return
    if "--sequence" in sys.argv:      # the last N launches in start order: name, duration, grid (one vocoder pass is ~140 launches)
        n = int(sys.argv[sys.argv.index("--sequence") + 1])
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
        wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
        rows = db.execute(f"select name, (end - start) / 1e3, {gx}, {wx}, start from kernels order by start desc limit {n}").fetchall()[::-1]
        t0 = rows[0][4] if rows else 0
        print(f"{'#':>4} {'t_us':>10} {'dur_us':>9} {'wgs':>7}  kernel      (columns of `kernels`: {cols})")
        for i, (name, dur, g, w, st) in enumerate(rows):
            print(f"{i:>4} {(st - t0) / 1e3:>10.1f} {dur:>9.2f} {int(g) // max(1, int(w)):>7}  {short(name, 110)}")
        return
    rows = db.execute("""select name, count(*), sum(end - start) / 1e3, avg(end - start) / 1e3, min(end - start) / 1e3,
                         max(end - start) / 1e3 from kernels group by name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1.0
    print(f"{'kernel':<100} {'calls':>7} {'total_us':>12} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{short(name, 100):<100} {n:>7} {tot:>12.1f} {avg:>9.2f} {mn:>9.2f} {mx:>9.2f} {100 * tot / total:>6.2f}")
    print(f"{'TOTAL':<100} {sum(r[1] for r in rows):>7} {total:>12.1f}")


def has(db, table, col):
    try:
        return col in [r[1] for r in db.execute(f"pragma table_info({table})")]
    except sqlite3.Error:
        return False


def pmc_fallback(db):
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    raise SystemExit(f"unexpected counters_collection schema: {cols}")


def short(name, n=70):
    name = name.replace("void ", "").replace("fdx::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


if __name__ == "__main__":
    main()
