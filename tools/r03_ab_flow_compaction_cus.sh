#!/bin/bash
# A/B: CUs left free by a flow batch's tile kernel, for the compaction flows (the scatter kernels of the other batch can use them)
for rep in 1 2; do
for k in -1 32 64 96; do
  echo "spare=$k (-1 = shipped plan)"
  if [ $k -lt 0 ]; then env X=1 timeout -k 10 200 python3 tools/path_bench.py --workload ${1:-C2} --iters 20 --paths offsets32_flow,spans32_flow,features32_flow 2>&1 | grep flow | cut -c1-140
  else LATOK_AB_SPARE_CUS=$k timeout -k 10 200 python3 tools/path_bench.py --workload ${1:-C2} --iters 20 --paths offsets32_flow,spans32_flow,features32_flow 2>&1 | grep flow | cut -c1-140; fi
done; done
