#!/bin/bash
# SQ counters of the fused edge-chain kernel (eager launches: rocprofv3 --pmc profiles a replayed hipGraph node by node at
# ~15 ms per dispatch -- slow enough to run into gpurun's limit, not a hang: profiles/r02_pmc.md).  Usage: tools/pmc_chain.sh <mode> <out-prefix>
set -e
cd /tmp && export TMPDIR=/tmp
MODE=${1:-f16x3}
OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/pmc_chain_$MODE}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${OUT}_$tag -o pmc -- python3 $GRAFT_REPO_ROOT/tools/chain_bench.py --modes $MODE --eager --launches 3 --piece-sums > ${OUT}_$tag.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(out + "_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "egnn_edge_chain" not in k:
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:32s} {v / n[(k, c)]:.4g}  (avg of {n[(k, c)]} dispatches)")
PY
