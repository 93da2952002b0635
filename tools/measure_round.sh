#!/bin/bash
# A round's measurement batch (run on the GPU box through gpurun).  Everything lands under gpurun_out/measure_<tag>/ ; what is to
# be judged is copied into profiles/ afterwards (tools/collect_round.sh <tag>).  A step killed at its limit ends the batch.
#   tools/measure_round.sh <tag> [part]     tag = r04 ...; part = bench | paths | pathstats | pmc | narrow | all (default all)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:?round tag, e.g. r04}; shift
O=$R/gpurun_out/measure_$TAG
mkdir -p $O
cd $R
PART=${1:-all}
LOG=$O/log_$PART.txt
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
if [ "$PART" = "bench" ] || [ "$PART" = "all" ]; then
step bench_c2 300 python3 bench.py --gpus 1 --steps 20 --warmup 5
step bench_c3 300 python3 bench.py --workload C3 --steps 20 --warmup 3 --no-cpu-baseline
step bench_c4 400 python3 bench.py --workload C4 --steps 5 --warmup 2 --no-cpu-baseline
step bench_c5 400 python3 bench.py --workload C5 --steps 5 --warmup 2 --no-cpu-baseline
step bench_c2_serial 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --in-flight 1 --no-cpu-baseline
step bench_c2_shared_input 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --shared-input --no-cpu-baseline
step bench_n2_turns 300 python3 bench.py --gpus 2 --devices 0,0 --take-turns --steps 20 --warmup 5
step bench_n2_shared 300 python3 bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 --no-cpu-baseline
step bench_n4_c5_strong_turns 400 python3 bench.py --gpus 4 --devices 0,0,0,0 --take-turns --workload C5 --strings 400 --steps 5 --warmup 2 --no-cpu-baseline
step bench_n2_torchrun_gloo 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 20 --warmup 5 --take-turns --no-cpu-baseline
fi
if [ "$PART" = "paths" ] || [ "$PART" = "all" ]; then
step paths_c2 300 python3 tools/path_bench.py --workload C2 --iters 20 --cpu 100000
step paths_c3 300 python3 tools/path_bench.py --workload C3 --iters 10
step host_path 400 python3 tools/host_path_rate.py
step latency 120 python3 tools/latency_bench.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b /tmp/prof_p /tmp/prof_p3
# (1) the kernel itself: one batch at a time -- the profiler's duration of a launch is then the kernel's own time
rm -rf /tmp/prof_b /tmp/prof_bf
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 > $O/bench_c2_under_rocprof.json 2>/dev/null
echo "rocprof bench (in-flight 1) rc=$?" | tee -a $LOG
cp /tmp/prof_b/b_kernel_stats.csv $O/kernel_stats_bench_c2.csv 2>/dev/null || find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_bench_c2.csv \;
# (2) the default command (batch flow): tile kernels that overlap another one listed apart
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf -o b -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_flow_under_rocprof.json 2>/dev/null
echo "rocprof bench (default, flow) rc=$?" | tee -a $LOG
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_bf -name '*results.db' | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline (default: batch flow, two batches in flight); round $TAG" --split-overlap > $O/kernel_stats_bench_c2_flow.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o paths -- python3 $R/tools/path_bench.py --workload C2 --iters 5 --paths mask,offsets,offsets32,spans,spans32,features,features32,utf8_mask,utf8_offsets,utf8_spans,bytes_mask,bytes_offsets,bytes_spans,rules_mask,kind_mask,kind_offsets,kind_offsets32,kind_spans,kind_spans32 > /dev/null 2>&1
echo "rocprof paths c2 rc=$?" | tee -a $LOG
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_p -name '*results.db' | head -1) "rocprofv3 --kernel-trace --stats -- python3 tools/path_bench.py --workload C2 --iters 5 --paths <the blocking paths> (1 MI355X, C2 = 1 M ASCII strings; the flow paths are left out: overlapped launches carry queue time in their duration); round $TAG" > $O/paths_kernel_stats.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p3 -o paths -- python3 $R/tools/path_bench.py --workload C3 --iters 5 --paths mask,offsets,offsets32,spans,spans32,features,features32,utf8_mask,utf8_offsets,utf8_spans,bytes_mask,bytes_offsets,bytes_spans,rules_mask,kind_mask,kind_offsets,kind_offsets32,kind_spans,kind_spans32 > /dev/null 2>&1
echo "rocprof paths c3 rc=$?" | tee -a $LOG
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_p3 -name '*results.db' | head -1) "rocprofv3 --kernel-trace --stats -- python3 tools/path_bench.py --workload C3 --iters 5 --paths <the blocking paths> (1 MI355X, C3 = 1 M mixed-Unicode strings); round $TAG" > $O/paths_kernel_stats_c3.txt
cd $R
fi
if [ "$PART" = "pathstats" ]; then
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_p /tmp/prof_p3
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o paths -- python3 $R/tools/path_bench.py --workload C2 --iters 5 --paths mask,offsets,offsets32,spans,spans32,features,features32,utf8_mask,utf8_offsets,utf8_spans,bytes_mask,bytes_offsets,bytes_spans,rules_mask,kind_mask,kind_offsets,kind_offsets32,kind_spans,kind_spans32 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_p -name '*results.db' | head -1) "rocprofv3 --kernel-trace --stats -- python3 tools/path_bench.py --workload C2 --iters 5 --paths <the blocking paths> (1 MI355X, C2 = 1 M ASCII strings; the flow paths are left out: overlapped launches carry queue time in their duration); round $TAG" > $O/paths_kernel_stats.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p3 -o paths -- python3 $R/tools/path_bench.py --workload C3 --iters 5 --paths mask,offsets,offsets32,spans,spans32,features,features32,utf8_mask,utf8_offsets,utf8_spans,bytes_mask,bytes_offsets,bytes_spans,rules_mask,kind_mask,kind_offsets,kind_offsets32,kind_spans,kind_spans32 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_p3 -name '*results.db' | head -1) "rocprofv3 --kernel-trace --stats -- python3 tools/path_bench.py --workload C3 --iters 5 --paths <the blocking paths> (1 MI355X, C3 = 1 M mixed-Unicode strings); round $TAG" > $O/paths_kernel_stats_c3.txt
cd $R
fi
if [ "$PART" = "pmc" ] || [ "$PART" = "all" ]; then
cd /tmp && export TMPDIR=/tmp
# PMC: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md), for the bench workloads at their full sizes
pmc() {  # label, limit, then the python command
  local label=$1 lim=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${label}_$c
    timeout -k 10 $lim rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${label}_$c -o p --output-format csv -- python3 "$@" > /dev/null 2>&1
    local rc=$?
    echo "pmc $label $c rc=$rc" | tee -a $LOG
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $LOG; exit 1; fi
    find /tmp/pmc_${label}_$c -name '*counter_collection.csv' -exec cp {} $O/pmc_${label}_$c.csv \;
  done
}
pmc c2 300 $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0 --in-flight 1
pmc c3 300 $R/bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0 --in-flight 1
pmc c4 500 $R/bench.py --workload C4 --steps 3 --warmup 1 --no-cpu-baseline --sustain-s 0 --in-flight 1
pmc c5 500 $R/bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --sustain-s 0 --in-flight 1
pmc c2_flow 300 $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0
pmc c2_flow_shared 300 $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0 --shared-input
pmc c3_flow 300 $R/bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0
pmc paths 300 $R/tools/path_bench.py --workload C2 --iters 3 --paths bytes_mask,kind_mask,offsets32,spans32,features32
pmc paths_c3 300 $R/tools/path_bench.py --workload C3 --iters 3 --paths bytes_mask,utf8_mask,kind_mask
cd $R
fi
if [ "$PART" = "narrow" ]; then
# instruction / LDS counters of the narrow-input tile kernels (-> profiles/<tag>_pmc_narrow_kernels.txt by tools/pmc_narrow_profile.py <tag>)
for wl in C3 C2; do
bash $R/tools/pmc_pass.sh ${TAG}_bytes_$wl "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" -- tools/path_bench.py --workload $wl --iters 3 --paths bytes_mask,kind_mask,utf8_mask > $O/narrow_$wl.log 2>&1
echo "narrow $wl rc=$?" | tee -a $LOG
done
fi
echo "=== done $PART" | tee -a $LOG
