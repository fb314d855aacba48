#!/bin/bash
# Round profile collection (run on the GPU box from the repo root): bench lines, kernel traces, PMC passes.
# A heartbeat keeps gpurun's silence watchdog quiet during the long, silent rocprofv3 runs.
# PMC passes: rocprofv3 --pmc serialises EVERY dispatch (about 15 ms each on this pool).  Round 1 ran them over
# `bench.py --forward pytorch` at its defaults -- 2 x 1000 iterations x 41 graph nodes = 82 000 dispatches, 20 minutes --
# and was killed at gpurun's limit (profiles/r02_pmc.md).  They now run over a bounded number of EAGER launches
# (--no-graph --steps 20 --warmup 2; the fused sampler is one launch per trajectory anyway), and the EGNN edge chain has its
# own pass on tools/chain_bench.py --eager (tools/pmc_chain.sh).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
( while true; do date >> $O/heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB" EXIT
cd $R
python bench.py --workload C2 > $O/bench_c2_fused.json 2> $O/bench_c2_fused.err
python bench.py --workload C2 --forward pytorch --no-cpu-baseline > $O/bench_c2_pytorch.json 2> $O/bench_c2_pytorch.err
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --workload C5 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rm -f $O/pmc_summary.txt
for tag in c2_fused c2_pytorch c3; do
  case $tag in
    c2_fused) ARGS="--workload C2"; PMC_ARGS="--workload C2";;
    c2_pytorch) ARGS="--workload C2 --forward pytorch"; PMC_ARGS="--workload C2 --forward pytorch --no-graph --steps 20 --warmup 2";;
    c3) ARGS="--no-graph"; PMC_ARGS="";;
  esac
  rm -rf $O/trace_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py --no-cpu-baseline $ARGS > $O/trace_$tag.log 2>&1
  cp $(find $O/trace_$tag -name '*kernel_stats.csv' | head -1) $O/stats_$tag.csv
  rm -f $O/trace_$tag/*/*kernel_trace.csv
  echo "trace $tag done"
  if [ $tag = c3 ]; then $R/tools/pmc_chain.sh f16x3 $O/pmc_chain_f16x3 > $O/pmc_chain_f16x3.txt 2>&1; continue; fi
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${tag}_$CTR
    timeout -k 10 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $O/pmc_${tag}_$CTR -- python3 $R/bench.py --no-cpu-baseline $PMC_ARGS > $O/pmc_${tag}_$CTR.log 2>&1
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
