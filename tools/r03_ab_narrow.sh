#!/bin/bash
# same-box A/B of narrow-input variants on C2 (and C3 for the byte kernel): path_bench kind_mask / bytes_mask
set -u
O=gpurun_out/r03ab; mkdir -p $O
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/latok_amd/liblatok_hip_$v.so; [ "$v" = "head" ] && lib=$PWD/latok_amd/liblatok_hip.so
  LATOK_HIP_LIB=$lib timeout -k 10 200 python tools/path_bench.py --workload C2 --iters 30 --paths kind_mask,bytes_mask,kind_offsets32,mask 2>/dev/null | python3 -c "import sys,json; [print('$v', 'C2', json.loads(l)['path'], round(json.loads(l)['ms_per_call'],4)) for l in sys.stdin if l.startswith('{')]"
done
done | tee $O/ab_narrow.txt
