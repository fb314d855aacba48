#!/bin/bash
# the shipped kernel's instruction stream on random operands and on all-zero activations: wall time, shader cycles, clock
set -e
cd $GRAFT_REPO_ROOT
python tools/chain_bench.py --lib tools/_ablate/libmdx_clk.so --modes f16x3 --piece-sums --clocks > gpurun_out/zero_a.log 2>&1
python tools/chain_bench.py --lib tools/_ablate/libmdx_clk.so --modes f16x3 --piece-sums --clocks --zero-activations > gpurun_out/zero_b.log 2>&1
python tools/chain_bench.py --modes f16x3 --piece-sums --eager > gpurun_out/zero_c.log 2>&1
python tools/chain_bench.py --modes f16x3 --piece-sums --eager --zero-activations > gpurun_out/zero_d.log 2>&1
tail -n 2 gpurun_out/zero_a.log gpurun_out/zero_b.log gpurun_out/zero_c.log gpurun_out/zero_d.log
