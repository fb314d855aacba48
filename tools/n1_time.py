"""radius_graph_kernel<fill> at the C3 and C5 sizes (hipGraph-timed, as bench.py does)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
for name in ("C3", "C5"):
    w = bench.WORKLOADS[name]
    m = bench.time_radius_graph(w["batch"], w, dev)
    print(name, round(m["ms"] * 1e3, 2), "us", round(m["bytes"] / (m["ms"] * 1e-3) / 1e9, 1), "GB/s")
