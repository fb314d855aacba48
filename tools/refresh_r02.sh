#!/bin/bash
# Round-2 final numbers (run on the GPU box from the repo root; results in gpurun_out/, copied to profiles/ by hand):
# the bench lines of the other workloads, the end-to-end jobs, the 2-rank rehearsal on one GPU.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python bench.py --workload C2 > $O/r02_bench_c2.json 2> $O/r02_bench_c2.err
echo "c2 done"
python bench.py --workload C4 --no-cpu-baseline > $O/r02_bench_c4.json 2> $O/r02_bench_c4.err
echo "c4 done"
python bench.py --workload C5 --no-cpu-baseline > $O/r02_bench_c5.json 2> $O/r02_bench_c5.err
python bench.py --workload C5 --resampling 0 --no-cpu-baseline > $O/r02_bench_c5_r0.json 2> $O/r02_bench_c5_r0.err
echo "c5 done"
python tools/e2e_c3.py C3 --repeat=2 > $O/r02_e2e_c3.json 2> $O/r02_e2e_c3.err
python tools/e2e_c3.py C4 >> $O/r02_e2e_c3.json 2>> $O/r02_e2e_c3.err
python tools/e2e_c3.py C3 --precision=f32 >> $O/r02_e2e_c3.json 2>> $O/r02_e2e_c3.err
echo "e2e done"
python bench.py --gpus 2 --backend gloo --no-cpu-baseline > $O/r02_bench_2rank_gloo_c3.json 2> $O/r02_bench_2rank_gloo_c3.err
python bench.py --gpus 2 --backend gloo --workload C2 --no-cpu-baseline > $O/r02_bench_2rank_gloo_c2.json 2> $O/r02_bench_2rank_gloo_c2.err
echo "2-rank done"
