#!/bin/bash
# The launches of one C3 network forward, in order (eager run under the kernel trace).  -> gpurun_out/forward_sequence.txt
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_seq
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_seq -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --whole-job-budget-s 0 > $O/seq_bench.log 2>&1
python3 $R/tools/kernel_sequence.py /tmp/tr_seq > $O/forward_sequence.txt
cat $O/forward_sequence.txt
