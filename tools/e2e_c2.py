"""End-to-end check of the bench's extrapolation: one full C2 job -- generator.sample(1024): initialisation, the 1000
iterations, the status read -- timed on the wall clock, next to bench.py's K-iteration figure."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

device = torch.device("cuda:0")
w = bench.WORKLOADS["C2"]
gen, noise, sampling, net = bench.build_generator(w, device, 0, w["batch"], False)
gen.fused_score_network = True
times = []
with torch.no_grad():
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gen.sample(w["batch"], device)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
assert (out.A != w["num_atom_types"]).all() and torch.isfinite(out.X).all() and (out.X >= 0).all() and (out.X < 1).all()
best = min(times[1:])
print(json.dumps({"job": "C2 generator.sample(1024), 1000 iterations, end to end", "seconds": [round(t, 5) for t in times],
                  "structures_per_s_best_of_4": round(w["batch"] / best, 1)}))
