#!/bin/bash
# Round-5 profiles (GPU box): tools/profile_r05.sh  -> gpurun_out/r05_*
#   the launches of one C3 forward in order; per-kernel statistics of the graph-replayed bench command (primary workload only:
#   `--also-measured no`, so that a kernel's average is over launches of ONE size); SQ / traffic counters of the edge chain
#   (separate --pmc passes over eager launches; the program itself right after `--`); the default bench line; the
#   distribution report.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_seq /tmp/tr_stats
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_seq -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --whole-job-budget-s 0 --also-measured no > $O/r05_seq_bench.log 2>&1
python3 $R/tools/kernel_sequence.py /tmp/tr_seq > $O/r05_c3_forward_sequence.txt
echo "sequence done"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --whole-job-budget-s 0 --also-measured no > $O/r05_bench_c3_profiled.json 2> $O/r05_stats_bench.err
cp $(find /tmp/tr_stats -name '*kernel_stats.csv' | head -1) $O/r05_c3_kernel_stats.csv
echo "stats done"
$R/tools/pmc_chain.sh f16x3 $O/r05_pmc_chain_f16x3 > $O/r05_pmc_chain_f16x3.txt 2>&1
echo "pmc done"
cd $R
python bench.py --steps 20 --warmup 5 > $O/r05_bench_default.json 2> $O/r05_bench_default.err; echo default $?
python tools/distribution_report.py > $O/r05_distribution_report.txt 2> $O/r05_distribution_report.err; echo report $?
