#!/bin/bash
# Round-3 profiles of the default bench line (C3): the launches of one network forward in order, and the per-kernel statistics
# of the graph-replayed run.  Usage (GPU box): tools/profile_r03.sh   -> gpurun_out/r03_c3_forward_sequence.txt, r03_c3_kernel_stats.csv
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_seq /tmp/tr_stats
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_seq -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --whole-job-budget-s 0 > $O/r03_seq_bench.log 2>&1
python3 $R/tools/kernel_sequence.py /tmp/tr_seq > $O/r03_c3_forward_sequence.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --whole-job-budget-s 0 > $O/r03_stats_bench.log 2>&1
cp $(find /tmp/tr_stats -name '*kernel_stats.csv' | head -1) $O/r03_c3_kernel_stats.csv
tail -c 1500 $O/r03_stats_bench.log
head -14 $O/r03_c3_kernel_stats.csv | cut -c1-160
