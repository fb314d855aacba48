#!/bin/bash
# Round-3 bench lines of the other workloads (the driver's own run covers the default, C3)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python bench.py --workload C4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r03_bench_c4.json 2> $O/r03_bench_c4.err; echo C4 $? 
python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_bench_c5.json 2> $O/r03_bench_c5.err; echo C5 $?
python bench.py --workload C5 --steps 2 --warmup 1 --resampling 0 --batch 512 --no-cpu-baseline > $O/r03_bench_c5_b512_r0.json 2> $O/r03_bench_c5_b512.err; echo C5b512 $?
python bench.py --workload C2 --no-cpu-baseline > $O/r03_bench_c2.json 2> $O/r03_bench_c2.err; echo C2 $?
python bench.py --egnn-precision f16x3_32x32 --steps 10 --warmup 3 --no-cpu-baseline --whole-job-budget-s 0 > $O/r03_bench_c3_32x32.json 2> $O/r03_bench_c3_32x32.err; echo C3-32 $?
for f in c4 c5 c5_b512_r0 c2 c3_32x32; do python - <<PY
import json
d=json.load(open("$O/r03_bench_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["value_from"][:40], d["config"].get("peak_device_memory_bytes"), d["config"].get("hip_graph"), d["roofline"].get("avg_launch_us"))
PY
done
