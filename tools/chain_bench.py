"""Time the fused MFMA edge-chain kernel at a workload's shape (default C3: E ~ 819 k edges, H = 256, 4 message + 5
coordinate layers) in both arithmetic modes, next to the same chain as per-layer hipBLASLt calls (mdx_linear_act)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd import kernels  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=512 * 64)
ap.add_argument("--degree", type=int, default=25)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--modes", default="f32,f16x3,library")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, n_in, n_msg, n_crd = args.hidden, args.hidden, 4, 5
n_nodes, E = args.nodes, args.nodes * args.degree
lin0 = torch.nn.Linear(2 * n_in + 1, H).to(dev)
msg = [torch.nn.Linear(H, H).to(dev) for _ in range(n_msg)]
crd = [torch.nn.Linear(H, H).to(dev) for _ in range(n_crd)]
out = torch.nn.Linear(H, 1, bias=False).to(dev)
src = torch.arange(n_nodes, device=dev).repeat_interleave(args.degree)
dst = (src // 64) * 64 + torch.randint(0, 64, (E,), device=dev)
edges = torch.stack([src, dst], 1).contiguous()
proj = torch.randn(n_nodes, 2 * H, device=dev)
coord = torch.rand(n_nodes, 6, device=dev)
flops = 2.0 * E * H * H * (n_msg + n_crd)
res = {"edges": E, "hidden": H, "layers": n_msg + n_crd, "algorithmic_gflop": flops / 1e9}
with torch.no_grad():
    for mode in args.modes.split(","):
        if mode == "library":
            radial = torch.rand(E, device=dev)

            def launch():
                x = kernels.egnn_message_input(proj, edges, radial, lin0.bias, lin0.weight[:, 2 * n_in].contiguous())
                for layer in msg:
                    x = kernels.linear_act(x, layer.weight, layer.bias, True)
                for layer in crd:
                    x = kernels.linear_act(x, layer.weight, layer.bias, True)
                return x
        else:
            pack = kernels.EdgeChainPack(lin0, msg, crd, out, input_size=n_in, precision=mode)

            def launch(pack=pack):
                return kernels.egnn_edge_chain(pack, proj, coord, edges)
        ms = bench.time_launches(launch, dev, args.launches)
        res[mode] = {"ms": round(ms, 4), "algorithmic_tflops": round(flops / ms / 1e9, 2)}
print(json.dumps(res))
