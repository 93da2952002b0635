#!/bin/bash
# copy what is to be judged from gpurun_out/measure_<tag>/ (tools/measure_round.sh <tag>) into profiles/, named per round
#   tools/collect_round.sh <tag>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:?round tag, e.g. r04}
F=$R/gpurun_out/measure_$TAG
P=$R/profiles
for w in c2 c2_serial c2_shared_input c3 c4 c5 n2_turns n2_shared n4_c5_strong_turns n2_torchrun_gloo; do [ -f $F/bench_$w.out ] && grep '^{' $F/bench_$w.out | tail -1 > $P/${TAG}_bench_$w.json; done
[ -f $F/bench_c2_under_rocprof.json ] && grep '^{' $F/bench_c2_under_rocprof.json > $P/${TAG}_bench_c2_under_rocprof.json
[ -f $F/kernel_stats_bench_c2.csv ] && cp $F/kernel_stats_bench_c2.csv $P/${TAG}_kernel_stats_bench_c2.csv
[ -f $F/kernel_stats_bench_c2_flow.txt ] && cp $F/kernel_stats_bench_c2_flow.txt $P/${TAG}_kernel_stats_bench_c2_flow.txt
[ -f $F/bench_c2_flow_under_rocprof.json ] && grep '^{' $F/bench_c2_flow_under_rocprof.json > $P/${TAG}_bench_c2_flow_under_rocprof.json
[ -f $F/paths_kernel_stats.txt ] && cp $F/paths_kernel_stats.txt $P/${TAG}_paths_kernel_stats.txt
[ -f $F/paths_kernel_stats_c3.txt ] && cp $F/paths_kernel_stats_c3.txt $P/${TAG}_paths_kernel_stats_c3.txt
[ -f $F/paths_c2.out ] && cp $F/paths_c2.out $P/${TAG}_paths_c2.jsonl
[ -f $F/paths_c3.out ] && cp $F/paths_c3.out $P/${TAG}_paths_c3.jsonl
[ -f $F/host_path.out ] && cp $F/host_path.out $P/${TAG}_host_path_rate.jsonl
[ -f $F/latency.out ] && cp $F/latency.out $P/${TAG}_latency.txt
if [ -f $F/pmc_c2_FETCH_SIZE.csv ]; then
for l in c2 c3 c4 c5 c2_flow c2_flow_shared c3_flow paths paths_c3; do for c in FETCH_SIZE WRITE_SIZE; do [ -f $F/pmc_${l}_$c.csv ] && cp $F/pmc_${l}_$c.csv $P/${TAG}_pmc_${l}_$c.csv; done; done
python3 $R/tools/pmc_traffic_summary.py $P/${TAG}_pmc_summary.json \
  "bench:C2:128005325:1000000:$F/pmc_c2_FETCH_SIZE.csv:$F/pmc_c2_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0 --in-flight 1" \
  "bench:C3:255924408:1000000:$F/pmc_c3_FETCH_SIZE.csv:$F/pmc_c3_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0 --in-flight 1" \
  "bench:C4:12800261086:100000000:$F/pmc_c4_FETCH_SIZE.csv:$F/pmc_c4_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --workload C4 --steps 3 --warmup 1 --no-cpu-baseline --sustain-s 0 --in-flight 1" \
  "bench:C5:10000000000:10000:$F/pmc_c5_FETCH_SIZE.csv:$F/pmc_c5_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --sustain-s 0 --in-flight 1" \
  "bench|in_flight=2|distinct_inputs=1:C2:128005325:1000000:$F/pmc_c2_flow_FETCH_SIZE.csv:$F/pmc_c2_flow_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0 (the default command: batch flow, each slot its own input copy; a counter pass runs the dispatches one after the other)" \
  "bench|in_flight=2|distinct_inputs=0:C2:128005325:1000000:$F/pmc_c2_flow_shared_FETCH_SIZE.csv:$F/pmc_c2_flow_shared_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0 --shared-input" \
  "bench|in_flight=2|distinct_inputs=1:C3:255924408:1000000:$F/pmc_c3_flow_FETCH_SIZE.csv:$F/pmc_c3_flow_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0" \
  "paths:C2:128005325:1000000:$F/pmc_paths_FETCH_SIZE.csv:$F/pmc_paths_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/path_bench.py --workload C2 --iters 3 --paths bytes_mask,kind_mask,offsets32,spans32,features32" \
  "paths:C3:255924408:1000000:$F/pmc_paths_c3_FETCH_SIZE.csv:$F/pmc_paths_c3_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/path_bench.py --workload C3 --iters 3 --paths bytes_mask,utf8_mask,kind_mask"
sed -i "s/\"round\": 2/\"round\": \"$TAG\"/" $P/${TAG}_pmc_summary.json
fi
[ -f $R/gpurun_out/${TAG}_bytes_C3_pmc_summary.txt ] && python3 $R/tools/pmc_narrow_profile.py $TAG
