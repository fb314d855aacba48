"""Per-tile cycle profile of one edge tile of workgroup 0 (a -DMDX_CHAIN_STAMPS=1 build): the shader-clock distance between
consecutive run_tile entries (stamp 4), with the waits for the weight stream inside each (stamps 1 -> 2 wait, 2 -> 3 barrier).
    python tools/stamp_tiles.py tools/_ablate/libmdx_st1.so"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels  # noqa: E402

_hip.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, n_nodes, degree = 256, 512 * 64, 25
E = n_nodes * degree
lin0 = torch.nn.Linear(2 * H + 1, H).to(dev)
msg = [torch.nn.Linear(H, H).to(dev) for _ in range(4)]
crd = [torch.nn.Linear(H, H).to(dev) for _ in range(5)]
out = torch.nn.Linear(H, 1, bias=False).to(dev)
src = torch.arange(n_nodes, device=dev).repeat_interleave(degree)
dst = (src // 64) * 64 + torch.randint(0, 64, (E,), device=dev)
edges = torch.stack([src, dst], 1).contiguous()
proj = torch.randn(n_nodes, 2 * H, device=dev)
coord = torch.rand(n_nodes, 6, device=dev)
with torch.no_grad():
    pack = kernels.EdgeChainPack(lin0, msg, crd, out, input_size=H, precision="f16x3")
    stamps = torch.zeros(8192, dtype=torch.int32, device=dev)
    for _ in range(3):
        kernels.egnn_edge_chain(pack, proj, coord, edges, status=stamps, piece_sums=True)
    torch.cuda.synchronize()
raw = stamps.view(torch.int64).cpu().numpy()
seq = [(int(r >> 48), int(r & ((1 << 48) - 1))) for r in raw[:4000] if r >> 48]
# one edge tile = from a stamp 9 to the next
starts = [k for k, (i, _) in enumerate(seq) if i == 9]
a, b = starts[2], starts[3]            # the third edge tile of workgroup 0 (steady state)
tile = seq[a:b]
print("stamps in the edge tile:", len(tile), "cycles:", tile[-1][1] - tile[0][1] + 0)
rows, cur = [], None
for k, (i, t) in enumerate(tile):
    if i in (4, 9, 10, 30, 31):
        if cur:
            rows.append(cur)
        cur = {"id": i, "t": t, "wait": 0, "barrier": 0}
    elif i == 2 and cur:
        cur["wait"] += t - tile[k - 1][1]
    elif i == 3 and cur:
        cur["barrier"] += t - tile[k - 1][1]
rows.append(cur)
for k, r in enumerate(rows):
    nxt = rows[k + 1]["t"] if k + 1 < len(rows) else seq[b][1]
    print(f"{k:3d} id {r['id']:2d}  cycles {nxt - r['t']:6d}  of which vmcnt wait {r['wait']:5d}  barrier {r['barrier']:5d}")
