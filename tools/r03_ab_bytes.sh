#!/bin/bash
# same-box A/B of byte-space variants: path_bench bytes_mask on C3 and C2
set -u
O=gpurun_out/r03ab; mkdir -p $O
for v in "$@"; do
  for w in C3 C2; do
    LATOK_HIP_LIB=$PWD/latok_amd/liblatok_hip_$v.so timeout -k 10 200 python tools/path_bench.py --workload $w --iters 20 --paths bytes_mask 2>/dev/null | python3 -c "import sys,json; [print('$v', '$w', round(json.loads(l)['ms_per_call'],4)) for l in sys.stdin if l.startswith('{')]"
  done
done | tee $O/ab_bytes.txt
