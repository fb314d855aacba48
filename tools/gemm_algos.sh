#!/bin/bash
# builds and runs tools/gemm_algos.cpp on the GPU box: every hipBLASLt candidate for the EGNN edge GEMM (fp32, SiLU epilogue)
set -e
cd $GRAFT_REPO_ROOT/tools
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -o /tmp/gemm_algos gemm_algos.cpp -L/opt/rocm/lib -lhipblaslt
timeout -k 10 300 /tmp/gemm_algos 819200 256 256 0 > $GRAFT_REPO_ROOT/gpurun_out/gemm_algos.log 2>&1
timeout -k 10 400 /tmp/gemm_algos 819200 256 256 1 >> $GRAFT_REPO_ROOT/gpurun_out/gemm_algos.log 2>&1
tail -20 $GRAFT_REPO_ROOT/gpurun_out/gemm_algos.log
