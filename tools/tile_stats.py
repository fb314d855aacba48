"""Instruction accounting of the edge-chain kernel's steady-state tiles, from the compiler's assembly (no GPU needed).

    python tools/tile_stats.py [--source FILE.hip] [--kernel 256,1,2] [hipcc flags ...]

A tile = the instructions between two `sched_barrier` markers that contain 48 MFMAs (split-f16, H = 256).  Printed per tile:
instruction counts by class and a first-order ISSUE MODEL of one wavefront alone on its SIMD (MI355X_MICROARCH.md,
'vector-instruction ISSUE cost'): every instruction takes an issue slot -- 8 cycles for a transcendental or an MFMA (the
MFMA holds the vector issue for 8 of its 32 cycles), ~60 for a direct-to-LDS request, 4 for anything else, `s_nop N` N + 1
-- and an MFMA cannot start before the previous one has left the matrix pipe (32 cycles).  The model ignores operand
latencies and memory waits; it says whether the MFMAs CAN be back to back given where the compiler put everything else:
modeled cycles per tile against the 1536 of the MFMAs alone, and how the non-MFMA issue time is distributed over the
tile's 16 k-steps."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
source = os.path.join(ROOT, "diffusion_for_multi_scale_molecular_dynamics_amd", "csrc", "mdx_egnn_chain.hip")
kernel = "256,1,2"
flags = []
while args:
    a = args.pop(0)
    if a == "--source":
        source = args.pop(0)
    elif a == "--kernel":
        kernel = args.pop(0)
    else:
        flags.append(a)
H, prec, mode = kernel.split(",")
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "chain.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fno-fast-math", "-fno-gpu-flush-denormals-to-zero", "--cuda-device-only", "-S",
                           f"-I{ROOT}/include", f"-I{os.path.dirname(source)}",
                           f"-I{ROOT}/diffusion_for_multi_scale_molecular_dynamics_amd/csrc", "-o", asm, source] + flags,
                          stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
name = f"_ZN12_GLOBAL__N_122egnn_edge_chain_kernelILi{H}ELi{prec}ELi{mode}EEEvNS_9ChainArgsE"
start = [i for i, ln in enumerate(lines) if ln.startswith(name + ":")][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm")][0]
meta = {}
for ln in lines:
    m = re.match(r"\s+\.(vgpr_count|sgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", ln)
    if m:
        meta.setdefault(m.group(1), []).append(int(m.group(2)))
segs, cur = [], []
for ln in lines[start:end]:
    if "MDX_TILE_END" in ln:
        segs.append(cur)
        cur = []
        continue
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    m = re.match(r"([a-z_0-9]+)\s*(.*)", t)
    if m:
        cur.append((m.group(1), m.group(2)))
segs.append(cur)


n_mfma = {"0": 128, "1": 48, "2": 96}[prec]


def cost(op, arg):
    if op.startswith("v_mfma"):
        return 8
    if op in ("v_exp_f32_e32", "v_exp_f32_e64", "v_rcp_f32_e32", "v_rcp_f32_e64", "v_log_f32_e32", "v_sqrt_f32_e32"):
        return 8
    if op.startswith("global_load_lds"):
        return 60
    if op == "s_nop":
        return int(arg.split()[0]) + 1
    return 4


def last_tile(seg):
    """the instructions of a segment from just behind the (n_mfma + 1)-th MFMA from its end (a segment may open with other
    phases -- the first layer, the aggregation -- or be cut at the top of a loop)"""
    idx = [i for i, (op, _) in enumerate(seg) if op.startswith("v_mfma")]
    if len(idx) < n_mfma:
        return None
    return seg[idx[-n_mfma - 1] + 1:] if len(idx) > n_mfma else seg


tiles = [t for t in (last_tile(s) for s in segs) if t is not None]
print(f"kernel <{kernel}>: {len(tiles)} tiles of {n_mfma} MFMAs in the text; total instructions {end - start}")
rows = []
for s in tiles:
    c = collections.Counter()
    t, pipe_free, steps = 0, 0, []
    since = 0
    k = 0
    for op, arg in s:
        if op.startswith("v_mfma"):
            c["mfma"] += 1
            start_t = max(t, pipe_free)
            pipe_free = start_t + (16 if "16x16x32" in op else (64 if "32x32x2" in op else 32))
            t = start_t + 8
            k += 1
            if k % (n_mfma // 16) == 0:
                steps.append(since)
                since = 0
            continue
        w = cost(op, arg)
        t += w
        since += w
        if op.startswith("v_accvgpr"):
            c["acc_mov"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("ds_"):
            c["ds"] += 1
        elif op == "s_waitcnt":
            c["waitcnt"] += 1
        elif op == "s_nop":
            c["nop"] += 1
        elif op.startswith("global_load_lds"):
            c["dma"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        else:
            c["other"] += 1
    total = max(t, pipe_free)
    rows.append((len(s), c, total, steps))
keys = ["mfma", "valu", "acc_mov", "ds", "waitcnt", "nop", "salu", "dma", "other"]
print("tile  instr " + " ".join(f"{k:>7s}" for k in keys) + "   model cycles   non-MFMA issue cycles per k-step")
for i, (n, c, total, steps) in enumerate(rows):
    print(f"{i:4d} {n:6d} " + " ".join(f"{c[k]:7d}" for k in keys) + f"   {total:6d}         " + " ".join(f"{v:3d}" for v in steps))
body = rows[1:-1] if len(rows) > 4 else rows
import statistics
print(f"median over {len(rows)} tiles: instructions {statistics.median(r[0] for r in rows):.0f}, model cycles {statistics.median(r[2] for r in rows):.0f} "
      f"(MFMAs alone: {32 * n_mfma})")
print("registers:", {k: v for k, v in meta.items() if k in ("vgpr_spill_count", "sgpr_spill_count")} and
      {k: max(v) for k, v in meta.items()})
