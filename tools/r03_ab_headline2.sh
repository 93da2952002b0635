#!/bin/bash
set -u
for rep in 1 2 3; do for v in "$@"; do
  lib=$PWD/latok_amd/liblatok_hip_$v.so; [ "$v" = "head" ] && lib=$PWD/latok_amd/liblatok_hip.so
  LATOK_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('$v', 'value', round(l['value'],1), 'step_ms', round(l['ms_per_step'],4), 'events', round(l['ms_per_step_events'],4), 'kernel_ms', round(l['roofline']['kernel_ms'],4), 'sustained', round(l['sustained']['ms_per_step'],4))"
done; done
