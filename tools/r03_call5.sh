#!/bin/bash
set -u
O=gpurun_out/r03c5; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "rule or rules" > $O/pytest.out 2>&1; rc=$?; tail -25 $O/pytest.out; echo "pytest rc=$rc"
