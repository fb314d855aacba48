"""What bounds the per-node part of a C3 graph layer (round 5): launch durations of the node kernel
egnn_edge_chain_kernel<256, 2, 3> and of egnn_node_gather at 1, 2 and 4 tiles of 128 rows per CU, with and without the next
layer's projections, timed as hipGraph replays with HIP events (bench.py::time_launches).

    python tools/node_path_probe.py > gpurun_out/r5_node_path.json

t(k tiles per CU) = fixed + k * round:  `round` = what a tile costs when the kernel's tile loop is in steady state, `fixed` = what
one launch pays once (launch, table staging, ring prime, the first tile's exposed loads, the last tile's stores draining)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
net = bench.egnn_experiment(1).eval().to(dev)
layer0, layer1 = net.egnn.graph_layers[0], net.egnn.graph_layers[1]
H = 256
out = {}
with torch.no_grad():
    for projects in (True, False):
        pack = layer0._node_mlp_pack(layer1 if projects else None)
        assert pack is not None and pack.projects == projects
        for tiles_per_cu in (1, 2, 4):
            M = 256 * 128 * tiles_per_cu
            h, agg = torch.randn(M, H, device=dev), torch.randn(M, H, device=dev) * 0.1
            ms = bench.time_launches(lambda: kernels.node_mlp_rows(pack, h, True, agg=agg), dev, 20)
            out[f"node_mlp projects={projects} tiles_per_cu={tiles_per_cu}"] = round(ms * 1e3, 2)
    # one tile per workgroup on a quarter / half of the chip's CUs: is the fixed part contention between CUs or a serial latency?
    pack = layer0._node_mlp_pack(layer1)
    for tiles in (64, 128, 256):
        M = 128 * tiles
        h, agg = torch.randn(M, H, device=dev), torch.randn(M, H, device=dev) * 0.1
        ms = bench.time_launches(lambda: kernels.node_mlp_rows(pack, h, True, agg=agg), dev, 20)
        out[f"node_mlp projects=True, {tiles} workgroups of one tile"] = round(ms * 1e3, 2)
    word = torch.zeros(1, dtype=torch.int32, device=dev)
    ms = bench.time_launches(lambda: kernels.index_add(word, 1), dev, 200)
    out["a one-thread kernel in the same protocol (per-launch floor inside a hipGraph)"] = round(ms * 1e3, 2)
    # the gather at C3's shape: 32768 nodes, ~25 edges each (sorted), piece rows in the compact layout
    n_nodes, deg = 512 * 64, 25
    E = n_nodes * deg
    src = torch.arange(n_nodes, device=dev).repeat_interleave(deg)
    dst = (src // 64) * 64 + torch.randint(0, 64, (E,), device=dev)
    edges = torch.stack([src, dst], 1).contiguous()
    degree = torch.full((n_nodes,), deg, dtype=torch.int64, device=dev)
    offsets = (torch.cumsum(degree, 0) - degree).contiguous()
    rows = kernels.lib().mdx_egnn_piece_rows(E, n_nodes)
    pieces = torch.randn(rows, H, device=dev)
    scalar, coord = torch.randn(E, device=dev), torch.rand(n_nodes, 6, device=dev)
    ms = bench.time_launches(lambda: kernels.egnn_node_gather(pieces, E, offsets, degree, True, None, scalar, coord, edges, True), dev, 50)
    out["node_gather C3"] = round(ms * 1e3, 2)
    ms = bench.time_launches(lambda: kernels.egnn_coord_aggregate(scalar, coord, edges, offsets, degree, True), dev, 50)
    out["coord_aggregate alone C3"] = round(ms * 1e3, 2)
    ms = bench.time_launches(lambda: kernels.segment_combine(pieces, E, offsets, degree, True), dev, 50)
    out["segment_combine alone C3"] = round(ms * 1e3, 2)
    x = torch.randn(n_nodes, H, device=dev)
    y = torch.empty_like(x)
    ms = bench.time_launches(lambda: y.copy_(x), dev, 50)
    out["copy of one [n_nodes, 256] f32 matrix (33.5 MB read + 33.5 MB written)"] = round(ms * 1e3, 2)
for key in ("True", "False"):
    t1, t2, t4 = (out[f"node_mlp projects={key} tiles_per_cu={k}"] for k in (1, 2, 4))
    out[f"node_mlp projects={key}: steady-state round (t4 - t2) / 2"] = round((t4 - t2) / 2, 2)
    out[f"node_mlp projects={key}: fixed part t1 - round"] = round(t1 - (t4 - t2) / 2, 2)
print(json.dumps(out, indent=1))
