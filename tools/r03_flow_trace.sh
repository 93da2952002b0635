#!/bin/bash
# GPU box: kernel timeline of the flow A/B (which launches overlap)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
d=/tmp/kt_flow; rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace -d $d -o t -- python3 $R/tools/flow_ab.py ${1:-1000000} ${2:-0} 20 1 > $d.out 2> $d.err
echo rc=$?; tail -2 $d.out
db=$(find $d -name '*results.db' | head -1)
python3 $R/tools/rocpd_timeline.py $db 70 0 > $R/gpurun_out/flow_timeline_${2:-0}.txt
python3 $R/tools/rocpd_stats.py $db flow > $R/gpurun_out/flow_stats_${2:-0}.txt
