#!/usr/bin/env python3
"""Build profiles/<round>_pmc_summary.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (one counter per pass, as
MI355X_MICROARCH.md prescribes: the two do not fit one pass).  usage:
    pmc_traffic_summary.py out.json  label:workload:total_chars:n_strings:fetch.csv:write.csv:"command"  [...]
Per kernel of each run: average FETCH_SIZE / WRITE_SIZE (KiB) per launch and the HBM bytes derived from them with the
guide's gfx950 correction (FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads -> x2; WRITE_SIZE as is)."""
import csv
import json
import re
import sys
from collections import defaultdict


def averages(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and not r["Kernel_Name"].startswith("__amd_rocclr"):
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def short(name):
    m = re.match(r"(?:void )?(?:latok::)?([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    out = {"round": 2, "counters": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace, one counter per pass",
           "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced "
                         "streaming read (16 B/lane) -> x2; WRITE_SIZE taken as is; unit KiB; kernels with narrower "
                         "accesses (scatter, counts) are uncalibrated in absolute terms",
           "runs": []}
    for spec in sys.argv[2:]:
        label, workload, total_chars, n_str, fcsv, wcsv, command = spec.split(":", 6)
        label, *extras = label.split("|")          # "bench|in_flight=2|distinct_inputs=1": what bench.py's traffic_source matches on
        meta = {k: int(v) for k, v in (e.split("=") for e in extras)}
        f, w = averages(fcsv, "FETCH_SIZE"), averages(wcsv, "WRITE_SIZE")
        for k in sorted(set(f) | set(w)):
            fk, nf = f.get(k, (0.0, 0))
            wk, nw = w.get(k, (0.0, 0))
            rd, wr = int(2 * fk * 1024), int(wk * 1024)
            out["runs"].append({"label": label, "workload": workload, "total_chars": int(total_chars), "n_strings": int(n_str),
                                "kernel": short(k).split("<")[0], "kernel_full": short(k), "launches_averaged": [nf, nw],
                                "FETCH_SIZE_KB_avg": fk, "WRITE_SIZE_KB_avg": wk, "hbm_read_bytes_per_launch": rd,
                                "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "command": command, "launches": min(nf, nw), **meta})
    with open(sys.argv[1], "w") as fo:
        json.dump(out, fo, indent=1)
    for r in out["runs"]:
        print(f"{r['label']:10s} {r['kernel_full']:34s} read {r['hbm_read_bytes_per_launch'] / 1e6:10.1f} MB  written {r['hbm_write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
