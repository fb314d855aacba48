#!/bin/bash
# SQ counters of the persistent sampler kernel of the C2 workload (instruction mix and stall attribution); one small group per
# pass, each under its own timeout (with --pmc rocprofv3 serialises every dispatch: the workload is named explicitly -- the
# default of bench.py is the graph-replayed C3 line, whose thousands of graph nodes take minutes to profile this way)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -f $O/pmc_sq_summary.txt
i=0
for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT"; do
  i=$((i+1))
  OUT=$O/pmc_sq_$i
  rm -rf $OUT
  timeout -k 10 300 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --workload C2 --no-cpu-baseline > $O/pmc_sq_$i.log 2>&1 || { echo "group $i failed: $GROUP" >> $O/pmc_sq_summary.txt; tail -3 $O/pmc_sq_$i.log >> $O/pmc_sq_summary.txt; continue; }
  F=$(find $OUT -name '*counter_collection.csv' | head -1)
  python3 - "$F" <<'PY' >> $O/pmc_sq_summary.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "mlp_pc_sample_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k}: per launch {[round(x) for x in v]}")
PY
done
cat $O/pmc_sq_summary.txt
