"""A/B of node-kernel builds (egnn_edge_chain_kernel<256, 2, 3>) at the C3 shape, interleaved in one process:
    python tools/node_stagger_ab.py tree,tools/_ablate/libmdx_X.so[,...]
Per library: microseconds per launch (hipGraph replays, HIP events), median of the rounds; outputs bit-compared with the first."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels  # noqa: E402

dev = torch.device("cuda:0")
tree = _hip.LIB_PATH
handles, names = [], []
for entry in sys.argv[1].split(","):
    path = tree if entry == "tree" else os.path.abspath(entry)
    _hip._lib, _hip.LIB_PATH = None, path
    handles.append(_hip.lib())
    names.append("tree" if entry == "tree" else os.path.basename(path).replace("libmdx_", "").replace(".so", ""))
_hip._lib = handles[0]
torch.manual_seed(1)
net = bench.egnn_experiment(1).eval().to(dev)
layer0, layer1 = net.egnn.graph_layers[0], net.egnn.graph_layers[1]
times = {n: [] for n in names}
with torch.no_grad():
    pack = layer0._node_mlp_pack(layer1)
    M, H = 256 * 128, 256
    h, agg = torch.randn(M, H, device=dev), torch.randn(M, H, device=dev) * 0.1
    outs = []
    for k, name in enumerate(names):
        _hip._lib = handles[k]
        outs.append([t.clone() for t in kernels.node_mlp_rows(pack, h, True, agg=agg) if isinstance(t, torch.Tensor)])
    same = {names[k]: all(torch.equal(a, b) for a, b in zip(outs[k], outs[0])) for k in range(len(names))}
    for _ in range(5):
        for k, name in enumerate(names):
            _hip._lib = handles[k]
            times[name].append(bench.time_launches(lambda: kernels.node_mlp_rows(pack, h, True, agg=agg), dev, 20) * 1e3)
print(json.dumps({n: dict(us_median=round(statistics.median(t), 2), us_all=[round(x, 1) for x in t], identical=same[n]) for n, t in times.items()}, indent=1))
