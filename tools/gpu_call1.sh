#!/bin/bash
# one gpurun call: GPU tests, default bench, C4 / C5 at full size, 2-rank rehearsal on the one GPU.  A step that is
# killed at its limit (124 / 137) ends the call: nothing else is started on a possibly wedged GPU.
set -u
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 lim=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/call1.log
  timeout -k 10 "$lim" "$@" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/call1.log
  tail -3 "gpurun_out/$name.out" | cut -c1-600 | tee -a gpurun_out/call1.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a gpurun_out/call1.log; exit 1; fi
  return 0
}
: > gpurun_out/call1.log
step r02_pytest 850 python -m pytest tests -m gpu -x -q
step r02_bench_c2 300 python bench.py
step r02_bench_c4 400 python bench.py --workload C4 --steps 5 --warmup 2 --no-cpu-baseline
step r02_bench_c5 400 python bench.py --workload C5 --steps 5 --warmup 2 --no-cpu-baseline
step r02_bench_2rank_turns 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dist-backend gloo --device 0 --take-turns
step r02_bench_2rank_shared 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --dist-backend gloo --device 0
echo "=== done" | tee -a gpurun_out/call1.log
