#!/bin/bash
# HBM traffic of the two graph-build kernels (egnn_graph_mask_kernel, egnn_graph_emit_kernel) at a BASELINE shape: separate
# rocprofv3 --pmc passes over eager launches (FETCH_SIZE, WRITE_SIZE; units and the gfx950 correction per MI355X_MICROARCH.md:
# 1 KiB units... reported raw here, converted in profiles/traffic_graph_r05.json).  Usage: tools/pmc_graph.sh <C3|C5> <out-prefix>
set -e
cd /tmp && export TMPDIR=/tmp
SHAPE=${1:-C3}
OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/pmc_graph_$SHAPE}
for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${OUT}_$tag -o pmc -- python3 $GRAFT_REPO_ROOT/tools/graph_build_probe.py --eager $SHAPE --launches 5 > ${OUT}_$tag.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(out + "_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "egnn_graph" not in k:
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:32s} {v / n[(k, c)]:.6g}  (avg of {n[(k, c)]} dispatches)")
PY
