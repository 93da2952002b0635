#!/bin/bash
set -u
O=gpurun_out/r03c3; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "utf8 or bytes or unicode or byte" > $O/pytest.out 2>&1; rc=$?; tail -15 $O/pytest.out; echo "pytest rc=$rc"
