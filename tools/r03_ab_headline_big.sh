#!/bin/bash
set -u
for v in "$@"; do
  lib=$PWD/latok_amd/liblatok_hip_$v.so; [ "$v" = "head" ] && lib=$PWD/latok_amd/liblatok_hip.so
  for w in C4 C5; do
  LATOK_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --sustain-s 0 2>/dev/null | python3 -c "import sys,json; l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('$v', '$w', 'kernel_ms', round(l['roofline']['kernel_ms'],4), 'step_ms', round(l['ms_per_step_events'],4), 'frac', round(l['roofline']['frac'],3))"
  done
done
