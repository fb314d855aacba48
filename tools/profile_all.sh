#!/bin/bash
# Round profile collection (run on the GPU box from the repo root): bench lines, kernel traces, PMC passes.
# A heartbeat keeps gpurun's silence watchdog quiet during the long, silent rocprofv3 runs.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
( while true; do date >> $O/heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB" EXIT
cd $R
python bench.py > $O/bench_c2_fused.json 2> $O/bench_c2_fused.err
python bench.py --forward pytorch --no-cpu-baseline > $O/bench_c2_pytorch.json 2> $O/bench_c2_pytorch.err
python bench.py --workload C3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --workload C5 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rm -f $O/pmc_summary.txt
for tag in c2_fused c2_pytorch c3; do
  case $tag in
    c2_fused) ARGS="";;
    c2_pytorch) ARGS="--forward pytorch";;
    c3) ARGS="--workload C3 --steps 1 --warmup 1";;
  esac
  rm -rf $O/trace_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py --no-cpu-baseline $ARGS > $O/trace_$tag.log 2>&1
  cp $(find $O/trace_$tag -name '*kernel_stats.csv' | head -1) $O/stats_$tag.csv
  rm -f $O/trace_$tag/*/*kernel_trace.csv
  echo "trace $tag done"
  if [ $tag = c3 ]; then continue; fi     # PMC passes serialise every kernel: skipped for the EGNN forward
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${tag}_$CTR
    rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $O/pmc_${tag}_$CTR -- python3 $R/bench.py --no-cpu-baseline $ARGS > $O/pmc_${tag}_$CTR.log 2>&1
    python3 - "$(find $O/pmc_${tag}_$CTR -name '*counter_collection.csv' | head -1)" $CTR $tag >> $O/pmc_summary.txt <<'PY'
import csv, sys, collections
f, ctr, tag = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != ctr: continue
    k = r["Kernel_Name"]
    for key in ("mlp_pc_sample_kernel", "pc_step_kernel", "radius_graph_kernel", "fill_time_sigma", "repaint_rows"):
        if key in k:
            acc[key][0] += 1; acc[key][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{tag} {ctr} {k}: launches {n}, mean per launch {v/n:.3f} KB")
PY
    rm -rf $O/pmc_${tag}_$CTR
  done
  echo "pmc $tag done"
done
echo "profiles done"
