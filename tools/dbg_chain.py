import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
from test_egnn_chain_gpu import _chain_reference, _rel_l2
cuda = torch.device("cuda:0")
H, n_msg, n_crd, n_nodes, n_in, D = 32, 1, 1, 40, 24, 6
torch.manual_seed(1)
lin0 = torch.nn.Linear(2 * n_in + 1, H); msg = [torch.nn.Linear(H, H) for _ in range(n_msg)]
crd = [torch.nn.Linear(H, H) for _ in range(n_crd)]; out = torch.nn.Linear(H, 1, bias=False)
g = torch.Generator().manual_seed(2)
src = torch.repeat_interleave(torch.arange(n_nodes), torch.full((n_nodes,), 5)); E = src.numel()
edges = torch.stack([src, torch.randint(0, n_nodes, (E,), generator=g)], 1)
h = torch.randn(n_nodes, n_in, generator=g); coord = torch.rand(n_nodes, D, generator=g)
want_m, want_s = _chain_reference(lin0, msg, crd, out, n_in, h, coord, edges)
mods = [m.to(cuda) for m in [lin0] + msg + crd + [out]]
for prec in ("f32", "f16x3"):
    pack = kernels.EdgeChainPack(mods[0], mods[1:2], mods[2:3], mods[3], input_size=n_in, precision=prec)
    print(prec, "exponents", pack.exponents.tolist(), "max|W|", [float(m.weight.abs().max()) for m in mods[1:]])
    w = mods[0].weight.detach()
    proj = torch.nn.functional.linear(h.to(cuda), torch.cat([w[:, :n_in], w[:, n_in:2 * n_in]], 0)).contiguous()
    st = torch.zeros(1, dtype=torch.int32, device=cuda)
    gm, gs = kernels.egnn_edge_chain(pack, proj, coord.to(cuda).contiguous(), edges.to(cuda), status=st)
    torch.cuda.synchronize()
    print(" status", int(st.item()), "msg err", _rel_l2(gm, want_m), "head err", _rel_l2(gs, want_s))
    print(" got m[0,:6]", gm[0, :6].tolist()); print(" want     ", want_m[0, :6].tolist())
    print(" got s[:4]", gs[:4].tolist(), "want", want_s[:4].tolist())
    r = (gm.double().cpu() / want_m)
    print(" ratio stats", float(r.median()), float(r.min()), float(r.max()))
