"""Build-time check of csrc/mdx_egnn_chain.hip (run by the Makefile on the compiler's -Rpass-analysis=kernel-resource-usage
remarks of that translation unit).

The production-size piece-sums instantiations egnn_edge_chain_kernel<256, PREC, 2, false> issue their weight-stream requests
without the guard wait states (mdx_egnn_chain.hip, issue_piece): a scalar register restored from a vector register (v_readlane of a spilled SGPR)
right in front of such a request would be read as its address too early.  That cannot happen while those kernels spill no
scalar register -- which is what this script enforces, so that another compiler version or flag set fails the BUILD instead
of faulting on the GPU.  It also refuses private-segment (scratch) use in them: a spill in the hot loop is a 2x slowdown."""
import re
import sys

text = open(sys.argv[1]).read()
bad, seen = [], 0
for block in re.split(r"remark: Function Name: ", text)[1:]:
    name = block.split()[0]
    m = re.search(r"egnn_edge_chain_kernelILi(\d+)ELi(\d)ELi(\d)ELb(\d)E", name)
    if not m or m.group(3) != "2":
        continue
    if m.group(4) == "1":
        # the attention instantiations (ATT) always use the guarded request form and are allowed to spill: the gate needs
        # registers of its own while both operand sets are live, and an E_GCL with attention is not a benchmarked shape
        continue
    seen += 1
    field = lambda key: int(re.search(key + r":\s*(\d+)", block).group(1))      # noqa: E731
    sgpr_spill, scratch = field(r"SGPRs Spill"), field(r"ScratchSize \[bytes/lane\]")
    if (sgpr_spill and m.group(1) == "256") or scratch:      # (only <256, PREC, 2> use the unguarded request form)
        bad.append(f"<{m.group(1)},{m.group(2)},2>: SGPR spills {sgpr_spill}, scratch {scratch} B/lane")
if seen < 8:
    sys.exit(f"check_chain_resources: expected the 8 piece-sums instantiations in the remarks, found {seen}")
if bad:
    sys.exit("check_chain_resources: a piece-sums instantiation spills -- build with -DMDX_CHAIN_GUARD_ALL or fix the spill:\n  "
             + "\n  ".join(bad))
print(f"check_chain_resources: {seen} piece-sums instantiations, no scalar-register spills, no scratch")
