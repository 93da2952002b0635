#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one or more passes).  usage: pmc_summary.py a.csv b.csv ..."""
import csv
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for path in sys.argv[1:]:
        seen = set()
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if k.startswith("__amd_rocclr"):
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                key = (k, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(acc):
        d = sorted(dur[k])
        print(f"{k[:100]}   launches {len(d)}  median_us {d[len(d) // 2]:.1f}")
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:28s} {sum(v) / len(v):16.1f}   (n={len(v)})")


if __name__ == "__main__":
    main()
