#!/bin/bash
set -u
O=gpurun_out/r03c2; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "utf8 or bytes or unicode or byte" > $O/pytest.out 2>&1; rc=$?; tail -5 $O/pytest.out; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python tools/path_bench.py --workload C3 --iters 10 --paths mask,bytes_mask,utf8_mask,bytes_offsets,utf8_offsets > $O/paths_c3.jsonl 2> $O/paths_c3.err; echo "paths c3 rc=$?"; cut -c1-230 $O/paths_c3.jsonl
timeout -k 10 300 python tools/path_bench.py --workload C2 --iters 10 --paths mask,bytes_mask,utf8_mask,kind_mask > $O/paths_c2.jsonl 2> $O/paths_c2.err; echo "paths c2 rc=$?"; cut -c1-230 $O/paths_c2.jsonl
