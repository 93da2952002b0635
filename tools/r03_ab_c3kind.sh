#!/bin/bash
set -u
for rep in 1 2; do for v in "$@"; do
  lib=$PWD/latok_amd/liblatok_hip_$v.so; [ "$v" = "head" ] && lib=$PWD/latok_amd/liblatok_hip.so
  LATOK_HIP_LIB=$lib timeout -k 10 200 python tools/path_bench.py --workload C3 --iters 20 --paths kind_mask,kind_offsets32 2>/dev/null | python3 -c "import sys,json; [print('$v', 'C3', json.loads(l)['path'], round(json.loads(l)['ms_per_call'],4)) for l in sys.stdin if l.startswith('{')]"
done; done
