#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
MODE=${1:-f16x3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2_$MODE
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1 || true
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${OUT}_$tag -o pmc -- python3 $GRAFT_REPO_ROOT/tools/chain_bench.py --modes $MODE --eager --launches 3 > ${OUT}_$tag.log 2>&1 || tail -5 ${OUT}_$tag.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(out + "_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "egnn_edge_chain" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print(f"  {c:32s} {v / n[(k, c)]:.4g}  (avg of {n[(k, c)]} dispatches)")
PY
rm -rf ${OUT}_*
