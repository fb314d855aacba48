#!/bin/bash
# Round-4 profiles (GPU box): tools/profile_r04.sh  -> gpurun_out/r04_*
#   the launches of one C3 forward in order; per-kernel statistics of the graph-replayed default bench command; SQ / traffic
#   counters of the edge chain (separate --pmc passes over eager launches; the program itself right after `--`); the bench
#   lines of the other workloads; the distribution report.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_seq /tmp/tr_stats
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_seq -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --whole-job-budget-s 0 > $O/r04_seq_bench.log 2>&1
python3 $R/tools/kernel_sequence.py /tmp/tr_seq > $O/r04_c3_forward_sequence.txt
echo "sequence done"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --whole-job-budget-s 0 > $O/r04_bench_c3_profiled.json 2> $O/r04_stats_bench.err
cp $(find /tmp/tr_stats -name '*kernel_stats.csv' | head -1) $O/r04_c3_kernel_stats.csv
echo "stats done"
$R/tools/pmc_chain.sh f16x3 $O/r04_pmc_chain_f16x3 > $O/r04_pmc_chain_f16x3.txt 2>&1
echo "pmc done"
cd $R
python bench.py --workload C4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_bench_c4.json 2> $O/r04_bench_c4.err; echo C4 $?
python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/r04_bench_c5.json 2> $O/r04_bench_c5.err; echo C5 $?
python bench.py --workload C5 --steps 3 --warmup 1 --resampling 0 --no-cpu-baseline > $O/r04_bench_c5_r0.json 2> $O/r04_bench_c5_r0.err; echo C5r0 $?
python bench.py --workload C2 > $O/r04_bench_c2.json 2> $O/r04_bench_c2.err; echo C2 $?
python bench.py > $O/r04_bench_c3.json 2> $O/r04_bench_c3.err; echo C3 $?
python tools/distribution_report.py > $O/r04_distribution_report.txt 2> $O/r04_distribution_report.err; echo report $?
for f in c3 c4 c5 c5_r0 c2; do python - <<PY
import json
d=json.loads(open("$O/r04_bench_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d["value_from"][:40], d["config"].get("peak_device_memory_bytes"), d["roofline"].get("avg_launch_us"), d["roofline"].get("frac"), d["config"].get("f16_range_fallbacks"))
PY
done
