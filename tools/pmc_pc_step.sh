#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_pcstep_$CTR
  timeout -k 10 120 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $O/pmc_pcstep_$CTR -- python3 $R/tools/pmc_pc_step.py > $O/pmc_pcstep_$CTR.log 2>&1
  python3 - "$(find $O/pmc_pcstep_$CTR -name '*counter_collection.csv' | head -1)" $CTR <<'PY'
import csv, sys
f, ctr = sys.argv[1:3]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr and "pc_step_kernel" in r["Kernel_Name"]]
pred, corr = vals[0::2], vals[1::2]
print(f"{ctr} pc_step_kernel<8,2>: predictor mean {sum(pred)/len(pred):.1f} KB, corrector mean {sum(corr)/len(corr):.1f} KB ({len(vals)} launches)")
PY
  rm -rf $O/pmc_pcstep_$CTR
done
