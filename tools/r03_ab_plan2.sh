#!/bin/bash
run() { echo "$1"; shift; env "$@" timeout -k 10 120 python3 tools/flow_ab.py 1000000 0 100 2 2>&1 | grep round | cut -c10-; }
for rep in 1 2 3; do
run "flow on all CUs (old)" LATOK_AB_SPARE_CUS=0
run "default (216 CUs planned, seg 145)" X=1
run "spare 32 (224, seg 140)" LATOK_AB_SPARE_CUS=32
run "seg 144 (218 segs)" LATOK_AB_SEG_TILES=144
run "plan 2 (nearest multiple of 12)" LATOK_AB_PLAN=2
run "seg 156 (201 segs)" LATOK_AB_SEG_TILES=156
done
