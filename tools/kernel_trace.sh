#!/bin/bash
# Dev / measurement tool (GPU box): per-kernel durations of a python command (rocprofv3 --kernel-trace --stats).
#   tools/kernel_trace.sh <out_name> <script.py> [args]   -> gpurun_out/<out_name>_kernel_stats.txt
set -u
out=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
script=$R/$1; shift
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
d=/tmp/kt_$out
rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d -o t -- python3 "$script" "$@" > $d.out 2> $d.err
rc=$?
if [ $rc -ne 0 ]; then echo "rc=$rc"; tail -3 $d.err; fi
db=$(find $d -name '*results.db' | head -1)
python3 "$R/tools/rocpd_stats.py" "$db" "rocprofv3 --kernel-trace --stats -- python3 $(basename $script) $*" > "$R/gpurun_out/${out}_kernel_stats.txt"
cat "$R/gpurun_out/${out}_kernel_stats.txt"
