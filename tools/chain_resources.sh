#!/bin/bash
# Register / spill listing of every egnn_edge_chain_kernel instantiation (cross-compiles: no GPU needed).  Fails if a
# production-size piece-sums instantiation (<256, PREC, 2>) has scalar-register spills: those skip the guard wait states in
# front of their weight-stream requests (csrc/mdx_egnn_chain.hip, issue_piece).  The BUILD runs the same check on the
# compiler's remarks (csrc/Makefile + csrc/check_chain_resources.py); this script is the human-readable listing.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero \
    --cuda-device-only -S -I$R/include -o $T/chain.s $R/diffusion_for_multi_scale_molecular_dynamics_amd/csrc/mdx_egnn_chain.hip 2>/dev/null
python3 - $T/chain.s <<'PY'
import re, sys
entries, cur = [], None
for line in open(sys.argv[1]):
    m = re.match(r"\s+(- )?\.(\w+):\s+(\S+)", line)
    if not m:
        continue
    if m.group(1) and m.group(2) == "agpr_count":       # first key of a kernel's metadata entry
        cur = {}
        entries.append(cur)
    if cur is not None:
        cur[m.group(2)] = m.group(3)
bad = 0
print(f"{'instantiation <H,PREC,MODE>':30s} vgpr+agpr  sgpr  scratch B  sgpr spills  vgpr spills")
for e in entries:
    m = re.search(r"egnn_edge_chain_kernelILi(\d+)ELi(\d)ELi(\d)E", e.get("name", ""))
    if not m:
        continue
    H, prec, mode = m.groups()
    print(f"<{H},{prec},{mode}>{'':24s} {e['vgpr_count']:>8s} {e['sgpr_count']:>5s} {e['private_segment_fixed_size']:>10s} "
          f"{e['sgpr_spill_count']:>12s} {e['vgpr_spill_count']:>12s}")
    bad += mode == "2" and H == "256" and e["sgpr_spill_count"] != "0"
if bad:
    sys.exit("a piece-sums instantiation spills scalar registers: give it the guarded request form")
print("piece-sums instantiations: no scalar-register spills")
PY
rm -rf $T
