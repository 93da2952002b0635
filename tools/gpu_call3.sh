#!/bin/bash
set -u
mkdir -p gpurun_out
LOG=gpurun_out/call3.log
: > $LOG
step() {
  local name=$1 lim=$2; shift 2
  echo "=== $name" | tee -a $LOG
  timeout -k 10 "$lim" "$@" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"
  local rc=$?
  echo "rc=$rc" | tee -a $LOG
  tail -6 "gpurun_out/$name.out" | cut -c1-300 | tee -a $LOG
  if [ $rc -ne 0 ]; then tail -25 "gpurun_out/$name.out" | cut -c1-300 | tee -a $LOG; tail -5 "gpurun_out/$name.err" | cut -c1-300 | tee -a $LOG; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $LOG; exit 1; fi
  return 0
}
step r02_tests3 850 python -m pytest tests -m gpu -x -q
bash tools/gpu_ab_paths.sh mask,offsets,spans,features,bytes_offsets,kind_offsets,kind_spans C2 20 ab tree
cat gpurun_out/ab_paths.log >> $LOG
echo "=== done" | tee -a $LOG
