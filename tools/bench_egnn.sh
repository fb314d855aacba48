#!/bin/bash
# bench lines of the EGNN workloads (run on the GPU box from the repo root)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python bench.py --workload C3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --workload C5 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
tail -n 1 $O/bench_c3.json $O/bench_c4.json $O/bench_c5.json | cut -c1-260
