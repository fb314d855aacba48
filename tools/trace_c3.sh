#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
( while true; do date >> $O/heartbeat.txt; sleep 45; done ) &
HB=$!
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 $R/bench.py --no-cpu-baseline --workload C3 --steps 1 --warmup 1 > $O/trace_c3.log 2>&1 || true
kill $HB
cp $(find $O/trace_c3 -name '*kernel_stats.csv' | head -1) $O/stats_c3.csv
rm -rf $O/trace_c3/*/*kernel_trace.csv
head -12 $O/stats_c3.csv | cut -c1-200
