#!/bin/bash
# Round-2 profile of the DEFAULT bench command (C3): rocprofv3 kernel trace + stats, and -- in separate passes, on eager
# launches (tools/chain_bench.py --eager; never over a replayed hipGraph, see profiles/r02_pmc.md) -- the PMC counters of the
# dominant kernel.  Run from the repo root on the GPU box; results land in gpurun_out/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_r02_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02_c3 -- python3 $R/bench.py --no-graph > $O/r02_bench_c3_profiled.json 2> $O/r02_bench_c3_profiled.err
find $O/prof_r02_c3 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/r02_c3_kernel_stats.csv
rm -rf $O/prof_r02_c3
python3 $R/bench.py > $O/r02_bench_c3.json 2> $O/r02_bench_c3.err
$R/tools/pmc_chain.sh f16x3 $O/pmc_chain_f16x3 > $O/r02_pmc_chain_f16x3.txt 2>&1
$R/tools/pmc_chain.sh f32 $O/pmc_chain_f32 > $O/r02_pmc_chain_f32.txt 2>&1
rm -rf $O/pmc_chain_*
