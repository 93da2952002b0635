#!/bin/bash
set -u
mkdir -p gpurun_out
LOG=gpurun_out/call2.log
: > $LOG
step() {
  local name=$1 lim=$2; shift 2
  echo "=== $name" | tee -a $LOG
  timeout -k 10 "$lim" "$@" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"
  local rc=$?
  echo "rc=$rc" | tee -a $LOG
  tail -4 "gpurun_out/$name.out" | cut -c1-400 | tee -a $LOG
  if [ $rc -ne 0 ]; then tail -5 "gpurun_out/$name.err" | cut -c1-400 | tee -a $LOG; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $LOG; exit 1; fi
  return 0
}
step r02_ctx_tests 600 python -m pytest tests/test_gpu_contexts.py tests/test_gpu_parity.py -m gpu -x -q
for v in "" _nobw _noro _coop; do
  export LATOK_HIP_LIB=$PWD/latok_amd/liblatok_hip$v.so
  step ab_c2$v 200 python tools/quick_bench.py 1000000 0 40
  step ab_c4s$v 200 python tools/quick_bench.py 12500000 0 10
done
unset LATOK_HIP_LIB
step ab_c2_again 200 python tools/quick_bench.py 1000000 0 40
echo "=== done" | tee -a $LOG
