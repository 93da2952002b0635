#!/bin/bash
# round 3, first GPU call: the GPU suite on the changed library + every way bench.py can be started
set -u
O=gpurun_out/r03c1; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest.out 2>&1; rc=$?; tail -3 $O/pytest.out; echo "pytest rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
run() { local n=$1; shift; timeout -k 10 300 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$? $(cut -c1-200 $O/$n.json)"; tail -2 $O/$n.err; }
run bench_n1 python3 bench.py --gpus 1 --steps 20 --warmup 5
run bench_n2_turns python3 bench.py --gpus 2 --devices 0,0 --take-turns --steps 20 --warmup 5
run bench_n2_shared python3 bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 --no-cpu-baseline
run bench_n4_turns python3 bench.py --gpus 4 --devices 0,0,0,0 --take-turns --steps 20 --warmup 5 --no-cpu-baseline
run bench_n2_refused python3 bench.py --gpus 2 --steps 5
run bench_torchrun_gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --take-turns --no-cpu-baseline
run bench_torchrun_nccl python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 --take-turns --no-cpu-baseline --dist-backend nccl
