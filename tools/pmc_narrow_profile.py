#!/usr/bin/env python3
"""Dev tool: profiles/<tag>_pmc_narrow_kernels.txt from the per-workload summaries tools/pmc_pass.sh leaves under gpurun_out/
(<tag>_bytes_C3_pmc_summary.txt, <tag>_bytes_C2_pmc_summary.txt: tools/measure_round.sh <tag> narrow), with VALU instructions per
tile for the tile kernels (SQ_INSTS_VALU / tiles of the workload).

usage: tools/pmc_narrow_profile.py <tag>
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILES = {"C3": {"4": 115148, "6": 62482}, "C2": {"4": 31252, "5": 31252}}   # tiles of 4096 units: bytes / UCS-2 units / Latin-1 units


def main():
    tag = sys.argv[1]
    out = [f"# round {tag}: PMC view of the narrow-input tile kernels and k_lead_compress (tools/measure_round.sh {tag} narrow = tools/pmc_pass.sh:",
           "# rocprofv3 --pmc <set> --kernel-trace, one counter set per pass; path_bench.py --iters 3 --paths bytes_mask,kind_mask,utf8_mask).",
           "# Tiles: C3 115 148 (bytes) / 62 482 (UCS-2 units), C2 31 252.  VALU per tile = SQ_INSTS_VALU / tiles.  Byte space on C3: round 3 ~2 650 VALU",
           "# per tile (VALU busy 95 % of the kernel); round 4: 2 077 (commit 7eeb285, 70 191 855 LDS bank conflicts per launch), 1 729 with the 6-bit",
           "# table and grouped lookups (36 731 218), then the figures below.", ""]
    for wl in ("C3", "C2"):
        out.append("## " + wl)
        cur = None
        for line in open(os.path.join(ROOT, "gpurun_out", f"{tag}_bytes_{wl}_pmc_summary.txt")):
            line = line.rstrip("\n")
            m = re.match(r"void latok::k_tiles_main<(\d+), false", line)
            if line and not line.startswith(" "):
                cur = m.group(1) if m else None
            out.append(line)
            m2 = re.match(r"\s+SQ_INSTS_VALU\s+([\d.]+)", line)
            if m2 and cur in TILES[wl]:
                out.append(f"      -> {float(m2.group(1)) / TILES[wl][cur]:8.0f} VALU per tile")
        out.append("")
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_narrow_kernels.txt")
    with open(path, "w") as f:
        f.write("\n".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
