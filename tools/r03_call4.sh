#!/bin/bash
set -u
O=gpurun_out/r03c4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_contexts.py tests/test_gpu_parity.py -m gpu -x -q -k "resident or pool or shard or contexts or example" > $O/pytest.out 2>&1; rc=$?; tail -15 $O/pytest.out; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python tools/host_path_rate.py > $O/host_path_rate.jsonl 2> $O/host_path_rate.err; echo "host_path rc=$?"; grep resident $O/host_path_rate.jsonl | cut -c1-220
