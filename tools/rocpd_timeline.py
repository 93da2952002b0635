#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 rocpd database: start / end relative to the first one listed, the
queue / stream they ran on.  usage: tools/rocpd_timeline.py <results.db> [n=40] [skip_from_end=0]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else None)
    sel = "name, start, end" + (", " + qcol if qcol else "") + (", " + scol if scol else "")
    rows = c.execute(f"select {sel} from kernels order by start").fetchall()
    rows = rows[len(rows) - skip - n: len(rows) - skip]
    t0 = rows[0][1]
    print("# columns of `kernels`: " + " ".join(cols))
    for r in rows:
        extra = " ".join(str(x) for x in r[3:])
        print(f"{(r[1] - t0) / 1e3:10.2f} {(r[2] - t0) / 1e3:10.2f} {(r[2] - r[1]) / 1e3:8.2f}  {extra:12s} {r[0][:40]}")


if __name__ == "__main__":
    main()
