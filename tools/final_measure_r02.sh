#!/bin/bash
# Round-2 measurement batch (run on the GPU box through gpurun).  Everything lands under gpurun_out/final2/ ; what is to
# be judged is copied into profiles/ afterwards (tools/collect_r02.sh).  A step killed at its limit ends the batch.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final2
mkdir -p $O
cd $R
LOG=$O/log.txt
: > $LOG
step() {
  local name=$1 lim=$2; shift 2
  echo "=== $name" | tee -a $LOG
  timeout -k 10 "$lim" "$@" > "$O/$name.out" 2> "$O/$name.err"
  local rc=$?
  echo "rc=$rc" | tee -a $LOG
  tail -2 "$O/$name.out" | cut -c1-250 | tee -a $LOG
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $LOG; exit 1; fi
  return 0
}
step bench_c2 300 python bench.py --gpus 1 --steps 20 --warmup 3
step bench_c3 300 python bench.py --workload C3 --steps 20 --warmup 3 --no-cpu-baseline
step bench_c4 400 python bench.py --workload C4 --steps 5 --warmup 2 --no-cpu-baseline
step bench_c5 400 python bench.py --workload C5 --steps 5 --warmup 2 --no-cpu-baseline
step paths_c2 300 python tools/path_bench.py --workload C2 --iters 20 --cpu 100000
step paths_c3 300 python tools/path_bench.py --workload C3 --iters 10
step host_path 300 python tools/host_path_rate.py
step latency 120 python tools/latency_bench.py
step sweep 200 python tools/batch_size_sweep.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b /tmp/prof_p
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c2_under_rocprof.json 2>/dev/null
echo "rocprof bench rc=$?" | tee -a $LOG
cp /tmp/prof_b/b_kernel_stats.csv $O/kernel_stats_bench_c2.csv 2>/dev/null || find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_bench_c2.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o paths -- python3 $R/tools/path_bench.py --workload C2 --iters 5 > /dev/null 2>&1
echo "rocprof paths rc=$?" | tee -a $LOG
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_p -name '*results.db' | head -1) "rocprofv3 --kernel-trace --stats -- python3 tools/path_bench.py --workload C2 --iters 5 (1 MI355X, C2 = 1 M ASCII strings); round 2" > $O/paths_kernel_stats.txt
# PMC: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md), for the bench workloads and the other kernels
pmc() {  # label, then the python command
  local label=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${label}_$c
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${label}_$c -o p --output-format csv -- python3 "$@" > /dev/null 2>&1
    echo "pmc $label $c rc=$?" | tee -a $LOG
    find /tmp/pmc_${label}_$c -name '*counter_collection.csv' -exec cp {} $O/pmc_${label}_$c.csv \;
  done
}
pmc c2 $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0
pmc c3 $R/bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0
pmc paths $R/tools/path_bench.py --workload C2 --iters 3 --paths bytes_mask,kind_mask,offsets32,spans32,features32
cd $R
echo "=== done" | tee -a $LOG
