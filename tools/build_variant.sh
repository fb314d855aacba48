#!/bin/bash
# Build a variant of the library for A/B runs on one box: tools/build_variant.sh NAME [extra hipcc flags for the chain unit]
#   -> tools/_ablate/libmdx_NAME.so  (the chain translation unit recompiled with the flags; the other objects as built in csrc/)
# `tools/build_variant.sh NAME --source FILE.hip [flags]` compiles FILE.hip in place of csrc/mdx_egnn_chain.hip.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/diffusion_for_multi_scale_molecular_dynamics_amd/csrc
name=$1; shift
src=$C/mdx_egnn_chain.hip
if [ "$1" == "--source" ]; then src=$2; shift 2; fi
mkdir -p $R/tools/_ablate
make -s -C $C mdx_hip.o mdx_egnn.o mdx_egnn_chain_att.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden \
    -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function -I$C "$@" -c -o $R/tools/_ablate/chain_$name.o $src
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_ablate/libmdx_$name.so $C/mdx_hip.o $C/mdx_egnn.o \
    $C/mdx_egnn_chain_att.o $R/tools/_ablate/chain_$name.o
echo "built tools/_ablate/libmdx_$name.so"
