#!/bin/bash
# Round-end measurement batch (run on the GPU box through gpurun): tests, smoke, bench lines, rocprof kernel stats, PMC passes.
# Everything lands under gpurun_out/final/ ; copy what is to be judged into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -20 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err || { tail $O/bench_c2.err; exit 1; }
: > $O/bench_all.jsonl
for w in C2 C3 C5; do timeout -k 10 600 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline >> $O/bench_all.jsonl 2>> $O/bench_all.err || exit 1; done
timeout -k 10 600 python bench.py --workload C2 --strings 12500000 --steps 10 --warmup 2 --no-cpu-baseline >> $O/bench_all.jsonl 2>> $O/bench_all.err || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b /tmp/pmc_f /tmp/pmc_w
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c2_under_rocprof.json 2>/dev/null || exit 1
cp /tmp/prof_b/b_kernel_stats.csv $O/kernel_stats_bench_c2.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o f --output-format csv -- python3 $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -o w --output-format csv -- python3 $R/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
cp /tmp/pmc_f/f_counter_collection.csv $O/pmc_fetch_size.csv
cp /tmp/pmc_w/w_counter_collection.csv $O/pmc_write_size.csv
cd $R
timeout -k 10 300 python tools/path_bench.py --workload C2 --cpu 100000 > $O/paths_c2.jsonl 2> $O/paths_c2.err || exit 1
timeout -k 10 300 python tools/path_bench.py --workload C3 > $O/paths_c3.jsonl 2> $O/paths_c3.err || exit 1
cd /tmp && rm -rf /tmp/prof_p && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o paths -- python3 $R/tools/path_bench.py --workload C2 --iters 5 > /dev/null 2>&1 || exit 1
python3 $R/tools/rocpd_stats.py /tmp/prof_p/paths_results.db "rocprofv3 --kernel-trace --stats -- python3 tools/path_bench.py --workload C2 --iters 5 (1 MI355X, C2 = 1 M ASCII strings); end of round 1" > $O/paths_kernel_stats.txt
echo "final measurement done"
head -c 600 $O/bench_c2.json
