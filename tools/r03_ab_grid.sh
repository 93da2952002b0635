#!/bin/bash
# same-box A/B: flow batches planned in several rounds of segments -- grid = the planned CUs (head) vs all CUs (ab)
for rep in 1 2; do for v in ab head; do
  lib=$PWD/latok_amd/liblatok_hip_ab.so; [ "$v" = "head" ] && lib=$PWD/latok_amd/liblatok_hip.so
  for n in 1000000 6000000 10000000; do
    it=100; [ $n -ge 3000000 ] && it=20
    echo "$v n=$n $(LATOK_HIP_LIB=$lib timeout -k 10 200 python3 tools/flow_ab.py $n 0 $it 2 2>&1 | grep round | tail -1 | cut -c10-)"
  done
done; done
