"""One full C3 job end to end: generator.sample(512) with the EGNN (4 x 256, rc 7.5), 1000 iterations x (1 predictor +
2 correctors), wall clock (graph capture and warm-up included); checks the result's properties.  ~40 s.
    python tools/e2e_c3.py [C3|C4] [--no-graph] [--precision=f32]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

device = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
w = bench.WORKLOADS[name]
use_graph = "--no-graph" not in sys.argv
gen, noise, sampling, net = bench.build_generator(w, device, 0, w["batch"], use_graph)
precision = next((a.split("=")[1] for a in sys.argv if a.startswith("--precision=")), "f16x3")
net.edge_chain_precision = precision
repeat = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--repeat=")), 1)
with torch.no_grad():
    for k in range(repeat):              # (--repeat=2: the second job has no first-call costs: library load, packs, capture)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gen.sample(w["batch"], device)
        torch.cuda.synchronize()
        seconds = time.perf_counter() - t0
        if k + 1 < repeat:
            print(json.dumps({"job": f"{name} call {k + 1} of {repeat}", "seconds": round(seconds, 2)}), flush=True)
assert (out.A != w["num_atom_types"]).all() and torch.isfinite(out.X).all() and (out.X >= 0).all() and (out.X < 1).all()
print(json.dumps({"job": f"{name} generator.sample({w['batch']}), {w['noise']['total_time_steps']} iterations, end to end",
                  "hip_graph": use_graph, "edge_chain": net.edge_chain_precision,
                  "seconds": round(seconds, 2), "structures_per_s": round(w["batch"] / seconds, 4)}))
