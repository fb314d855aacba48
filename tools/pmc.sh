#!/bin/bash
# PMC passes (one counter per pass, with --kernel-trace only) for the dominant kernel of a workload.
# usage: bash tools/pmc.sh <tag> <bench args...>
set -e
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
for CTR in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${CTR}
  rm -rf $OUT
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${CTR}.log 2>&1
  F=$(find $OUT -name '*counter_collection.csv' | head -1)
  python3 - "$F" $CTR <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != ctr: continue
    k = r["Kernel_Name"]
    for key in ("pc_step_kernel", "radius_graph_kernel", "fill_time_sigma", "repaint_rows", "coords_update"):
        if key in k:
            name = key + ("<fill>" if "true" in k or "<true>" in k else "<count>" if "radius" in key else "")
            acc[name][0] += 1; acc[name][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{ctr} {k}: launches {n}, mean {v/n:.3f} (counter units; x1024 = bytes if KB)")
PY
done
