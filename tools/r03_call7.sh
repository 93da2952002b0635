#!/bin/bash
set -u
O=gpurun_out/r03c7; mkdir -p $O
run() { local n=$1; shift; timeout -k 10 300 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$?"; tail -2 $O/$n.err; }
run n1_eager python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
LATOK_BENCH_GRAPH=1 run n1_graph python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run n2_threads_shared python3 bench.py --gpus 2 --devices 0,0 --launch threads --steps 20 --warmup 5 --no-cpu-baseline
LATOK_BENCH_GRAPH=0 run n2_threads_shared_eager python3 bench.py --gpus 2 --devices 0,0 --launch threads --steps 20 --warmup 5 --no-cpu-baseline
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c7/*.json")):
    t=[l for l in open(f) if l.startswith("{")]
    if not t: print(f, "no line"); continue
    l=json.loads(t[-1]); print(f.split("/")[-1], {k:l.get(k) for k in ["n_gpus","value","ms_per_step","ms_per_step_events","ms_per_rank","start_skew_us"]})
PY
