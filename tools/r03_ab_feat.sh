#!/bin/bash
for rep in 1 2 3; do
for v in ab head; do
  lib=$PWD/latok_amd/liblatok_hip_ab.so; [ "$v" = "head" ] && lib=$PWD/latok_amd/liblatok_hip.so
  LATOK_HIP_LIB=$lib timeout -k 10 200 python3 tools/path_bench.py --workload C2 --iters 20 --paths features32,rules_mask 2>/dev/null | python3 -c "import sys,json; [print('$v', 'C2', json.loads(l)['path'], round(json.loads(l)['ms_per_call'],4)) for l in sys.stdin if l.startswith('{')]"
  LATOK_HIP_LIB=$lib timeout -k 10 200 python3 tools/path_bench.py --workload C3 --iters 10 --paths features32,rules_mask 2>/dev/null | python3 -c "import sys,json; [print('$v', 'C3', json.loads(l)['path'], round(json.loads(l)['ms_per_call'],4)) for l in sys.stdin if l.startswith('{')]"
done; done
