#!/bin/bash
# A/B of the segment plan (whole wave rounds) and of the flow's CU share, per batch size, same box, alternating
for n in 300000 500000 800000 1000000 1500000 2500000 4000000; do
  for rep in 1 2; do
    echo "n_str=$n old (LATOK_AB_PLAN=0, flow on all CUs)"
    LATOK_AB_PLAN=0 LATOK_AB_SPARE_CUS=0 timeout -k 10 120 python3 tools/flow_ab.py $n 0 100 1 2>&1 | grep round
    echo "n_str=$n new"
    timeout -k 10 120 python3 tools/flow_ab.py $n 0 100 1 2>&1 | grep round
  done
done
echo "C3 old"; LATOK_AB_PLAN=0 LATOK_AB_SPARE_CUS=0 timeout -k 10 120 python3 tools/flow_ab.py 1000000 1 100 2 2>&1 | grep round
echo "C3 new"; timeout -k 10 120 python3 tools/flow_ab.py 1000000 1 100 2 2>&1 | grep round
