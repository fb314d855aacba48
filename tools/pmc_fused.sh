#!/bin/bash
# kernel trace + PMC passes (one counter per pass) of the default bench command (C2, fused persistent sampler)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_c2_fused
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2_fused -- python3 $R/bench.py --workload C2 --no-cpu-baseline > $O/prof_c2_fused.log 2>&1
find $O/prof_c2_fused -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/c2_fused_kernel_stats.csv
echo "kernel trace done" >> $O/heartbeat.txt
for CTR in FETCH_SIZE WRITE_SIZE; do
  OUT=$O/pmc_c2_fused_${CTR}
  rm -rf $OUT
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --workload C2 --no-cpu-baseline > $O/pmc_c2_fused_${CTR}.log 2>&1
  echo "$CTR done" >> $O/heartbeat.txt
  F=$(find $OUT -name '*counter_collection.csv' | head -1)
  python3 - "$F" $CTR <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/pmc_c2_fused_summary.txt
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != ctr: continue
    k = r["Kernel_Name"]
    for key in ("mlp_pc_sample_kernel", "pc_noise_fill_kernel"):
        if key in k: acc[key].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{ctr} {k}: launches {len(v)}, per launch {[round(x,1) for x in v]}, sum {sum(v):.1f} (KiB)")
PY
done
head -5 $O/c2_fused_kernel_stats.csv | cut -c1-200
