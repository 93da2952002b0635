#!/bin/bash
# copy what is to be judged from gpurun_out/final2/ (tools/final_measure_r02.sh) into profiles/, named per round
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
F=$R/gpurun_out/final2
P=$R/profiles
for w in c2 c3 c4 c5; do grep '^{' $F/bench_$w.out > $P/r02_bench_$w.json; done
grep '^{' $F/bench_c2_under_rocprof.json > $P/r02_bench_c2_under_rocprof.json
cp $F/kernel_stats_bench_c2.csv $P/r02_kernel_stats_bench_c2.csv
cp $F/paths_kernel_stats.txt $P/r02_paths_kernel_stats.txt
cp $F/paths_c2.out $P/r02_paths_c2.jsonl
cp $F/paths_c3.out $P/r02_paths_c3.jsonl
cp $F/host_path.out $P/r02_host_path_rate.jsonl
cp $F/latency.out $P/r02_latency_final.txt      # (r02_latency.txt keeps the before / after history of the round)
[ -f $F/sweep.out ] && cp $F/sweep.out $P/r02_batch_size_sweep_final.txt
for l in c2 c3 paths; do for c in FETCH_SIZE WRITE_SIZE; do cp $F/pmc_${l}_$c.csv $P/r02_pmc_${l}_$c.csv; done; done
python3 $R/tools/pmc_traffic_summary.py $P/r02_pmc_summary.json \
  "bench:C2:128005325:1000000:$F/pmc_c2_FETCH_SIZE.csv:$F/pmc_c2_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0" \
  "bench:C3:255924408:1000000:$F/pmc_c3_FETCH_SIZE.csv:$F/pmc_c3_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline --sustain-s 0" \
  "paths:C2:128005325:1000000:$F/pmc_paths_FETCH_SIZE.csv:$F/pmc_paths_WRITE_SIZE.csv:rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/path_bench.py --workload C2 --iters 3 --paths bytes_mask,kind_mask,offsets32,spans32,features32"
