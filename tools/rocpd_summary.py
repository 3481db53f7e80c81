#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace --stats) as text.

usage: tools/rocpd_summary.py <results.db> [--pmc]
Prints per-kernel calls / total / average / min / max duration (us), registers and LDS,
and, if counters were collected, per-kernel counter means.
"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-112s %6s %12s %11s %11s %11s %6s %5s %5s %5s %7s %8s %10s %5s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds", "scratch",
        "grid_x", "wg_x"))
    for r in rows:
        print("%-112s %6d %12.1f %11.2f %11.2f %11.2f %6.2f %5d %5d %5d %7d %8d %10d %5d" % (
            short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
            r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0, r[11] or 0, r[12] or 0))
    if "--pmc" in sys.argv:
        try:
            q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                 "group by kernel_name, counter_name")
            print("\ncounters (mean per dispatch)")
            for name, cname, val, n in cur.execute(q):
                print("%-90s %-28s %18.1f  (n=%d)" % (short(name, 90), cname, val, n))
        except sqlite3.Error as e:
            print("no counters:", e)


if __name__ == "__main__":
    main()
