#!/bin/bash
# round 3, batch flow: the N > 1 forms on the one GPU + the other workloads through the flow
set -u
O=gpurun_out/r03c8; mkdir -p $O
run() { local n=$1; shift; timeout -k 10 400 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$?"; tail -2 $O/$n.err; }
run n1 python3 bench.py --gpus 1 --steps 20 --warmup 5
run n1_serial python3 bench.py --gpus 1 --steps 20 --warmup 5 --in-flight 1 --no-cpu-baseline
run n2_procs_turns python3 bench.py --gpus 2 --devices 0,0 --take-turns --steps 20 --warmup 5 --no-cpu-baseline
run n2_procs_shared python3 bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 --no-cpu-baseline
run n2_threads_turns python3 bench.py --gpus 2 --devices 0,0 --take-turns --launch threads --steps 20 --warmup 5 --no-cpu-baseline
run n4_procs_turns python3 bench.py --gpus 4 --devices 0,0,0,0 --take-turns --steps 20 --warmup 5 --no-cpu-baseline
run n2_torchrun_gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --take-turns --device 0 --steps 20 --warmup 5 --no-cpu-baseline
run c3 python3 bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline
run c5 python3 bench.py --workload C5 --steps 10 --warmup 2 --no-cpu-baseline --sustain-s 0
run c4 python3 bench.py --workload C4 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c8/*.json")):
    t=[l for l in open(f) if l.startswith("{")]
    if not t: print(f, "no line"); continue
    l=json.loads(t[-1]); s=l.get("serial") or {}
    print(f.split("/")[-1], {k:l.get(k) for k in ["n_gpus","in_flight","value","value_projected","ms_per_step","ms_per_rank"]}, "serial", s.get("value"), s.get("ms_per_step"), s.get("ms_per_step_projected"), "frac", l["roofline"]["frac"], l["roofline"]["pipeline_frac"])
PY
