#!/usr/bin/env python3
"""Per-kernel summary (calls, total / avg / min / max duration in us, share) of a rocprofv3 rocpd database, the
table `rocprofv3 --kernel-trace --stats` would print.  usage: tools/rocpd_stats.py <results.db> [title]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                     "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    if len(sys.argv) > 2:
        print("# " + sys.argv[2])
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}")
    for r in rows:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]:10.1f} {r[3]:9.1f} {r[4]:8.1f} {r[5]:8.1f} {100 * r[2] / tot:6.1f}")


if __name__ == "__main__":
    main()
