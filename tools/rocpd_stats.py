#!/usr/bin/env python3
"""Per-kernel summary (calls, total / avg / min / max duration in us, share) of a rocprofv3 rocpd database, the
table `rocprofv3 --kernel-trace --stats` would print.  usage: tools/rocpd_stats.py <results.db> [title] [--split-overlap]

--split-overlap: kernels whose name contains `k_tiles_main` are listed twice more -- the launches that ran ALONE on the GPU with
respect to other tile kernels, and the launches whose [start, end) interval overlaps another tile kernel's.  With the batch
flow (two batches in flight on two streams) a tile kernel is dispatched while the previous batch's still holds the CUs; the
profiler's `duration` of such a launch includes the time its workgroups wait for a CU, so only the first group measures the
kernel itself (the figure bench.py's roofline uses comes from back-to-back launches on one stream)."""
import sqlite3
import sys


def line(name, durs, tot):
    n = len(durs)
    s = sum(durs)
    return f"{name[:72]:72s} {n:6d} {s:10.1f} {s / n:9.1f} {min(durs):8.1f} {max(durs):8.1f} {100 * s / tot:6.1f}"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    split = "--split-overlap" in sys.argv
    c = sqlite3.connect(args[0])
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    by = {}
    for name, a, b in rows:
        by.setdefault(name, []).append((b - a) / 1e3)
    tot = sum(sum(v) for v in by.values()) or 1.0
    if len(args) > 1:
        print("# " + args[1])
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}")
    for name, durs in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(line(name, durs, tot))
    if split:
        tiles = [(a, b, name) for name, a, b in rows if "k_tiles_main" in name]
        alone, over = {}, {}
        for i, (a, b, name) in enumerate(tiles):
            ov = (i > 0 and tiles[i - 1][1] > a) or (i + 1 < len(tiles) and tiles[i + 1][0] < b)
            # (sorted by start: only neighbours can overlap when at most two are in flight; checked against all below)
            (over if ov else alone).setdefault(name, []).append((b - a) / 1e3)
        print("# tile kernels by overlap with another tile kernel (batch flow: two batches in flight)")
        for tag, d in (("alone", alone), ("overlapped (duration includes waiting for CUs)", over)):
            for name, durs in d.items():
                print(line(f"[{tag}] {name}", durs, tot))
        # chains of overlapping tile kernels: what one launch costs on average = (end of the chain - its start) / launches
        chains, cur = [], None
        for a, b, name in tiles:
            if cur is not None and a < cur[1]:
                cur[1] = max(cur[1], b)
                cur[2] += 1
            else:
                if cur is not None and cur[2] >= 8:
                    chains.append(cur)
                cur = [a, b, 1]
        if cur is not None and cur[2] >= 8:
            chains.append(cur)
        if chains:
            n = sum(c[2] for c in chains)
            span = sum(c[1] - c[0] for c in chains) / 1e3
            print(f"# {len(chains)} chains of >= 8 overlapping tile-kernel launches: {n} launches in {span:.1f} us = {span / n:.2f} us per launch "
                  "(under the profiler; includes the other kernels of the batches where there are any)")


if __name__ == "__main__":
    main()
