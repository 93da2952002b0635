#!/bin/bash
set -u
O=gpurun_out/r03c6; mkdir -p $O
run() { local n=$1; shift; timeout -k 10 300 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$? $(cut -c1-160 $O/$n.json)"; tail -2 $O/$n.err; }
run n1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run n2_procs_turns python3 bench.py --gpus 2 --devices 0,0 --take-turns --steps 20 --warmup 5
run n2_procs_shared python3 bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 --no-cpu-baseline
run n4_procs_turns python3 bench.py --gpus 4 --devices 0,0,0,0 --take-turns --steps 20 --warmup 5 --no-cpu-baseline
run n2_threads_turns python3 bench.py --gpus 2 --devices 0,0 --take-turns --launch threads --steps 20 --warmup 5 --no-cpu-baseline
run n2_refused python3 bench.py --gpus 2 --steps 5
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c6/*.json")):
    t=[l for l in open(f) if l.startswith("{")]
    if not t: print(f, "no line"); continue
    l=json.loads(t[-1]); print(f.split("/")[-1], {k:l.get(k) for k in ["n_gpus","value","value_projected","ms_per_step","ms_per_rank","start_skew_us"]}, l["config"]["launch"][:40], "cpu" , "cpu_baseline" in l)
PY
