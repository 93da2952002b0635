#!/bin/bash
# Dev / measurement tool (GPU box): one rocprofv3 counter pass per counter SET over a python command, summarised per kernel.
#   tools/pmc_pass.sh <out_prefix> "<SET1 counters>" "<SET2 counters>" ... -- <script.py> [args]
# Counters in their own runs with --kernel-trace only (never with sys/hip/hsa tracing), the program itself after `--`.
set -u
out=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
: > "$R/gpurun_out/${out}_pmc.csvlist"
i=0
for s in "${sets[@]}"; do
  d=/tmp/pmc_${out}_$i
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $s --kernel-trace -d $d -o p --output-format csv -- python3 "$script" "$@" > /dev/null 2> $d.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass '$s' rc=$rc"; tail -3 $d.err; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; fi
  f=$(find $d -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" "$R/gpurun_out/${out}_pmc_$i.csv"; echo "$R/gpurun_out/${out}_pmc_$i.csv" >> "$R/gpurun_out/${out}_pmc.csvlist"; fi
  i=$((i+1))
done
python3 "$R/tools/pmc_summary.py" $(cat "$R/gpurun_out/${out}_pmc.csvlist") > "$R/gpurun_out/${out}_pmc_summary.txt"
cat "$R/gpurun_out/${out}_pmc_summary.txt"
