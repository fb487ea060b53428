#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel calls / total / mean / min / max duration, and PMC counter
sums per kernel when the run collected counters.  Usage: rocprof_summary.py <results.db> [> profiles/xxx.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 kernel-trace summary of {path}")
    print(f"{'kernel':<78} {'calls':>7} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{name[:78]:<78} {n:>7} {tot / 1e3:>11.1f} {avg / 1e3:>9.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} "
              f"{100 * tot / total:>6.2f}")
    try:
        pmc = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                          "group by kernel_name, counter_name order by kernel_name").fetchall()
    except sqlite3.Error as exc:  # schema differences between rocprofv3 builds
        pmc = []
        print(f"# (no PMC table: {exc})")
    if pmc:
        print("\n# PMC counters per kernel (sum over dispatches, mean per dispatch)")
        print(f"{'kernel':<60} {'counter':<32} {'n':>6} {'sum':>16} {'mean/dispatch':>16}")
        for name, ctr, n, s, a in pmc:
            print(f"{name[:60]:<60} {ctr:<32} {n:>6} {s:>16.1f} {a:>16.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
