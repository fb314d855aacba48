"""Time of the EGNN score network's graph build (mdx_egnn_radius_graph) in its two forms, at the BASELINE shapes.

    python tools/graph_build_probe.py [--out gpurun_out/graph_build.json]

Per shape: microseconds per call (launches captured back to back in a hipGraph, HIP events around the replay; bench.time_launches)
for  masks + emission (two launches)  and  count / scan / fill (three), the edges per atom, and the algorithmic bytes of
SURVEY 8(d) (12 B per atom read, 8 + 8 B per atom of offsets / counts written, 16 B per edge written) over the time of the call.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd import kernels  # noqa: E402

SHAPES = {"C3 (B 512, N 64, 10.86 A -> 16.5)": (512, 64, 10.86), "C5 (B 256, N 216, 16.29 A -> 16.5)": (256, 216, 16.29),
          "B 1024, N 8 (5.43 A -> 16.5)": (1024, 8, 5.43), "B 2048, N 64": (2048, 64, 10.86)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--launches", type=int, default=50)
    ap.add_argument("--eager", default=None, metavar="SHAPE",
                    help="for counter passes (rocprofv3 --pmc): --launches eager calls of the two-launch form at the shape whose name "
                         "starts with SHAPE (C3 / C5), nothing timed")
    args = ap.parse_args()
    device = torch.device("cuda:0")
    rc = 7.5
    report = {}
    for name, (B, N, box) in SHAPES.items():
        if args.eager and not name.startswith(args.eager):
            continue
        g = torch.Generator().manual_seed(B + N)
        x = torch.rand(B, N, 3, generator=g).to(device)
        lattice = torch.tensor([box, box, box, 0.0, 0.0, 0.0]).repeat(B, 1).to(device)
        capacity = B * N * (N - 1)
        status = torch.zeros(1, dtype=torch.int32, device=device)
        row = {}
        outs = {}
        if args.eager:
            for _ in range(args.launches):
                out = kernels.egnn_radius_graph(x, lattice, 2.2 * rc, rc, capacity, status=status, two_launches=True)
            torch.cuda.synchronize()
            E = int(out["n_edges"].item())
            print(json.dumps(dict(shape=name, edges=E, algorithmic_bytes=B * N * 28 + 16 * E, workspace_bytes_each_way=8 * (B * N * ((N + 63) // 64) + B))))
            continue
        for form, two in (("masks_emit", True), ("count_scan_fill", False)):
            outs[form] = kernels.egnn_radius_graph(x, lattice, 2.2 * rc, rc, capacity, status=status, two_launches=two)
            keep = []

            def launch():
                keep.append(kernels.egnn_radius_graph(x, lattice, 2.2 * rc, rc, capacity, status=status, two_launches=two))
                del keep[:-2]
            ms = bench.time_launches(launch, device, args.launches)
            E = int(outs[form]["n_edges"].item())
            nbytes = B * N * (12 + 16) + 16 * E
            row[form] = dict(us_per_call=round(ms * 1e3, 2), gb_per_s=round(nbytes / (ms * 1e-3) / 1e9, 1),
                             frac_of_8_tb_s=round(nbytes / (ms * 1e-3) / 8e12, 4))
        E = int(outs["masks_emit"]["n_edges"].item())
        same = all(torch.equal(outs["masks_emit"][k][:E] if k == "edges" else outs["masks_emit"][k],
                               outs["count_scan_fill"][k][:E] if k == "edges" else outs["count_scan_fill"][k])
                   for k in ("counts", "offsets", "edges", "n_edges"))
        row.update(edges=E, edges_per_atom=round(E / (B * N), 2), algorithmic_bytes=B * N * 28 + 16 * E, identical_outputs=same)
        report[name] = row
        print(name, json.dumps(row), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
