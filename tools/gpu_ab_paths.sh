#!/bin/bash
# usage: tools/gpu_ab_paths.sh "<paths>" <workload> <iters> variant1 variant2 ...   ("" = the in-tree library)
set -u
mkdir -p gpurun_out
LOG=gpurun_out/ab_paths.log
: > $LOG
paths=$1; wl=$2; iters=$3; shift 3
for v in "$@"; do
  if [ "$v" = "tree" ]; then unset LATOK_HIP_LIB; else export LATOK_HIP_LIB=$PWD/latok_amd/liblatok_hip_$v.so; fi
  echo "=== $v" | tee -a $LOG
  timeout -k 10 300 python tools/path_bench.py --workload $wl --iters $iters --paths $paths > gpurun_out/ab_paths_$v.jsonl 2> gpurun_out/ab_paths_$v.err
  rc=$?
  echo "rc=$rc" | tee -a $LOG
  python - <<PY | tee -a $LOG
import json
for l in open("gpurun_out/ab_paths_$v.jsonl"):
    d=json.loads(l)
    print("   %-14s %8.4f ms  %7.1f GB/s utf8  frac %.3f" % (d["path"], d["ms_per_call"], d["utf8_GBps"], d["frac_of_hbm_peak"]))
PY
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stop" | tee -a $LOG; exit 1; fi
done
