#!/bin/bash
# per-batch time of the mask path, blocking (serial) vs batch flow, over batch sizes (device resident, final library)
for m in 0 1; do for n in 50000 100000 300000 1000000 3000000 10000000; do
  it=100; [ $n -ge 3000000 ] && it=30
  echo "model=$m n_str=$n"; timeout -k 10 200 python3 tools/flow_ab.py $n $m $it 2 2>&1 | grep round | tail -1 | cut -c10-
done; done
