"""A/B of builds of the library IN ONE PROCESS, rounds interleaved (the guide's rule 24: separate invocations add
cross-process variance that looks like a kernel property; boxes differ by up to 12 % on MFMA-dense loops).

    python tools/ab_chain.py --libs tree,tools/_ablate/libmdx_X.so[,...] [--rounds 7] [--launches 4] [--piece-sums]

Every library gets its own handle, its own packed images and the same random operands (C3 shape by default: 786 432
edges, H = 256, 4 + 5 layers); one round = `launches` back-to-back launches of each variant in turn, timed with HIP events on
the launch stream after a warm-up; reported: median and minimum over rounds, microseconds per launch.  With --check the
outputs of every variant are compared with the first one's (max abs / rel-L2 difference)."""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True, help="comma-separated: 'tree' (csrc/libmdx_hip.so) or paths")
ap.add_argument("--nodes", type=int, default=512 * 64)
ap.add_argument("--degree", type=int, default=24)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--launches", type=int, default=4)
ap.add_argument("--mode", default="f16x3")
ap.add_argument("--n-msg", type=int, default=4)
ap.add_argument("--n-crd", type=int, default=5)
ap.add_argument("--rows-mode", action="store_true", help="messages as rows (default: piece sums, the product's mode)")
ap.add_argument("--check", action="store_true")
ap.add_argument("--any-abi", action="store_true", help="accept libraries of another MDX_ABI_VERSION (A/B against an older build)")
ap.add_argument("--warm-seconds", type=float, default=1.0, help="back-to-back launches before timing (clock settles)")
args = ap.parse_args()

dev = torch.device("cuda:0")
tree = _hip.LIB_PATH
names, handles, modes = [], [], []
for entry in args.libs.split(","):
    entry, _, mode = entry.partition(":")          # "lib[:mode]": e.g. tree:f16x3_32x32
    modes.append(mode or args.mode)
    path = tree if entry == "tree" else os.path.abspath(entry)
    _hip._lib, _hip.LIB_PATH = None, path
    if args.any_abi:          # (an older build of the library: its structs are prefixes of today's, some entry points absent)
        import ctypes

        class Tolerant:
            def __init__(self, lib):
                self.lib = lib

            def __getattr__(self, name):
                try:
                    return getattr(self.lib, name)
                except AttributeError:
                    return type("Absent", (), {})()

        old = ctypes.CDLL(path)
        _hip._declare(Tolerant(old))
        _hip._lib = old
    handles.append(_hip.lib())
    names.append(("tree" if entry == "tree" else os.path.basename(path).replace("libmdx_", "").replace(".so", "")) +
                 (":" + mode if mode else ""))

torch.manual_seed(0)
H, n_in = args.hidden, args.hidden
n_nodes, E = args.nodes, args.nodes * args.degree
lin0 = torch.nn.Linear(2 * n_in + 1, H).to(dev)
msg = [torch.nn.Linear(H, H).to(dev) for _ in range(args.n_msg)]
crd = [torch.nn.Linear(H, H).to(dev) for _ in range(args.n_crd)]
out = torch.nn.Linear(H, 1, bias=False).to(dev)
src = torch.arange(n_nodes, device=dev).repeat_interleave(args.degree)
dst = (src // 64) * 64 + torch.randint(0, 64, (E,), device=dev)
edges = torch.stack([src, dst], 1).contiguous()
proj = torch.randn(n_nodes, 2 * H, device=dev)
coord = torch.rand(n_nodes, 6, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)

packs = []
with torch.no_grad():
    for h, mode in zip(handles, modes):
        _hip._lib = h
        packs.append(kernels.EdgeChainPack(lin0, msg, crd, out, input_size=n_in, precision=mode))


def launch(k):
    _hip._lib = handles[k]
    return kernels.egnn_edge_chain(packs[k], proj, coord, edges, status=status, piece_sums=not args.rows_mode)


times = [[] for _ in handles]
with torch.no_grad():
    outs = [launch(k) for k in range(len(handles))]
    torch.cuda.synchronize()
    if args.check:
        for k in range(1, len(handles)):
            dm = (outs[k][1] - outs[0][1]).abs().max().item()
            rel = ((outs[k][1] - outs[0][1]).norm() / outs[0][1].norm()).item()
            same = [bool(torch.equal(outs[k][i], outs[0][i])) for i in range(2)]
            print(f"check {names[k]} vs {names[0]}: head max abs diff {dm:.3e}, rel-L2 {rel:.3e}; bit-identical messages / head: {same}")
    del outs
    import time
    t0 = time.time()
    while time.time() - t0 < args.warm_seconds:
        for k in range(len(handles)):
            launch(k)
        torch.cuda.synchronize()
    for r in range(args.rounds):
        order = list(range(len(handles)))
        if r % 2:
            order.reverse()
        for k in order:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.launches):
                launch(k)
            b.record()
            torch.cuda.synchronize()
            times[k].append(a.elapsed_time(b) * 1000.0 / args.launches)
res = {"edges": E, "hidden": H, "status": int(status.item())}
for k, name in enumerate(names):
    res[name] = {"median_us": round(statistics.median(times[k]), 1), "min_us": round(min(times[k]), 1),
                 "max_us": round(max(times[k]), 1)}
print(json.dumps(res))
