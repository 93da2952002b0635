#!/bin/bash
# the driver's round-end sequence, rehearsed: GPU tests, smoke, default bench (timed)
set -u
mkdir -p gpurun_out
timeout -k 10 850 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.out 2>&1; rc=$?; tail -3 gpurun_out/final_pytest.out; echo "pytest rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
s=$(date +%s); timeout -k 10 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; rc=$?; e=$(date +%s)
echo "bench rc=$rc seconds=$((e - s))"; cut -c1-400 gpurun_out/final_bench.json; tail -2 gpurun_out/final_bench.err
