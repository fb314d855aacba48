#!/bin/bash
# rocprofv3 kernel trace of the default bench command (run from the repo root on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT
shift_args="${@:2}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $shift_args > $GRAFT_REPO_ROOT/gpurun_out/prof_$1.log 2>&1
find $OUT -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_$1_kernel_stats.csv
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_$1.log
