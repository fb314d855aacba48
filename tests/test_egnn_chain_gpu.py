"""The hand-written MFMA edge-chain kernel (csrc/mdx_egnn_chain.hip) against fp64 references, through the C ABI."""
import os

import numpy as np
import math

import pytest
import torch

import nets

pytestmark = pytest.mark.gpu


def _rel_l2(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got.detach() - want.detach()).norm() / want.detach().norm().clamp(min=1e-300))


def _chain_reference(lin0, msg, crd, out, n_in, h, coord, edges):
    """E_GCL's per-edge part in fp64 exactly as models/egnn.py:136-200 composes it (concat -> Linear -> SiLU ...)."""
    f64 = torch.float64
    src, dst = edges[:, 0], edges[:, 1]
    diff = coord.to(f64)[src] - coord.to(f64)[dst]
    radial = (diff ** 2).sum(1, keepdim=True)
    x = torch.cat([h.to(f64)[src], h.to(f64)[dst], radial], dim=1)
    silu = torch.nn.functional.silu
    x = silu(x @ lin0.weight.to(f64).t() + lin0.bias.to(f64))
    for layer in msg:
        x = silu(x @ layer.weight.to(f64).t() + layer.bias.to(f64))
    y = x
    for layer in crd:
        y = silu(y @ layer.weight.to(f64).t() + layer.bias.to(f64))
    return x, (y @ out.weight.to(f64).t()).reshape(-1)


# tolerance per arithmetic mode: rel-L2 of the [E, H] messages and of the per-edge scalar against fp64
TOLERANCE = {"f32": 2e-6, "f16x3": 1e-5, "f16x3_32x32": 1e-5}
CHAIN_MODES = ["f32", "f16x3", "f16x3_32x32"]       # exact-f32 MFMA | split-f16 on 16x16x32 (default) | split-f16 on 32x32x16


@pytest.mark.parametrize("precision", CHAIN_MODES)
@pytest.mark.parametrize("H,n_msg,n_crd,n_nodes,deg", [(32, 1, 1, 40, 7), (64, 2, 3, 300, 11), (128, 3, 2, 500, 25),
                                                        (256, 4, 5, 1200, 25), (256, 1, 1, 3, 2)])
def test_edge_chain_against_fp64(cuda, precision, H, n_msg, n_crd, n_nodes, deg):
    """Random network, random node features, a ragged sorted edge list whose length is not a multiple of the 128-edge
    workgroup tile: messages [E,H] and the coordinate head's scalar against the fp64 evaluation of the reference's
    composition (asymmetric random weights: any transposed or permuted fragment map shows up as an O(1) error)."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    g = torch.Generator().manual_seed(1000 * H + n_nodes)
    n_in, D = 24, 6
    torch.manual_seed(H + n_msg)
    lin0 = torch.nn.Linear(2 * n_in + 1, H)
    msg = [torch.nn.Linear(H, H) for _ in range(n_msg)]
    crd = [torch.nn.Linear(H, H) for _ in range(n_crd)]
    out = torch.nn.Linear(H, 1, bias=False)
    for layer in msg + crd:                      # activations of order one through the whole chain (default init shrinks them)
        with torch.no_grad():
            layer.weight.mul_(1.7)
    degree = torch.randint(0, 2 * deg, (n_nodes,), generator=g)
    degree[0] = 0
    degree[-1] = max(int(degree[-1]), 6)
    src = torch.repeat_interleave(torch.arange(n_nodes), degree)
    E = int(src.numel())
    dst = torch.randint(0, n_nodes, (E,), generator=g)
    edges = torch.stack([src, dst], 1)
    h = torch.randn(n_nodes, n_in, generator=g)
    coord = torch.rand(n_nodes, D, generator=g) * 2 - 1
    want_m, want_s = _chain_reference(lin0, msg, crd, out, n_in, h, coord, edges)

    mods = [m.to(cuda) for m in [lin0] + msg + crd + [out]]
    lin0_d, msg_d, crd_d, out_d = mods[0], mods[1:1 + n_msg], mods[1 + n_msg:-1], mods[-1]
    assert kernels.EdgeChainPack.supported(lin0_d, msg_d, crd_d, out_d)
    pack = kernels.EdgeChainPack(lin0_d, msg_d, crd_d, out_d, input_size=n_in, precision=precision)
    w = lin0_d.weight.detach()
    proj = torch.nn.functional.linear(h.to(cuda), torch.cat([w[:, :n_in], w[:, n_in:2 * n_in]], 0)).contiguous()
    status = torch.zeros(1, dtype=torch.int32, device=cuda)
    coord_d, edges_d = coord.to(cuda).contiguous(), edges.to(cuda)
    got_m, got_s = kernels.egnn_edge_chain(pack, proj, coord_d, edges_d, status=status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert torch.isfinite(got_m).all() and torch.isfinite(got_s).all()
    tol = TOLERANCE[precision]
    err_m, err_s = _rel_l2(got_m, want_m), _rel_l2(got_s, want_s)
    assert err_m < tol and err_s < tol, (precision, H, err_m, err_s)
    # per-row check as well: no single edge (e.g. of the ragged last tile) may be off
    row_err = ((got_m.double().cpu() - want_m).norm(dim=1) / want_m.norm(dim=1).clamp(min=1e-30)).max()
    assert float(row_err) < 20 * tol, float(row_err)
    # a device-resident edge count smaller than the capacity: rows beyond it are not touched
    n_dev = torch.tensor([E - 5], dtype=torch.int64, device=cuda)
    m2 = torch.full((E, H), -7.0, device=cuda)
    s2 = torch.full((E,), -7.0, device=cuda)
    from diffusion_for_multi_scale_molecular_dynamics_amd._hip import check, lib, ptr, stream_handle
    import ctypes as C
    if E > 5:
        check(lib().mdx_egnn_edge_chain(C.byref(pack.c_struct), ptr(proj, torch.float32, "p"),
                                        ptr(coord_d, torch.float32, "c"), D,
                                        ptr(edges_d, torch.int64, "e"), E, ptr(n_dev, torch.int64, "n"),
                                        ptr(m2, torch.float32, "m"), ptr(s2, torch.float32, "s"), None, stream_handle()),
              "mdx_egnn_edge_chain")
        torch.cuda.synchronize()
        assert torch.equal(m2[:E - 5], got_m[:E - 5]) and torch.equal(s2[:E - 5], got_s[:E - 5])
        assert (m2[E - 5:] == -7.0).all() and (s2[E - 5:] == -7.0).all()


@pytest.mark.parametrize("precision", CHAIN_MODES)
@pytest.mark.parametrize("case", ["small_activations", "tiny_activations", "small_weights", "mixed_weights"])
def test_edge_chain_small_magnitudes_against_fp64(cuda, precision, case):
    """The UNDERFLOW side of the split-f16 mode.  f16's subnormal step is 2^-24, so an unscaled split v = hi + lo stops
    carrying 22 bits once |v| < 2^-3: default-initialised 256-wide layers (|W| <= 1/16) and activations of order 1e-3 are
    already there.  The kernel therefore keeps the weight image scaled by an exact per-layer power of two (largest
    |W| just under 2^14) and the carried activations by a fixed one (mdx_egnn_chain_t.activation_exponent), and undoes both
    in the epilogue -- exact, so O(1) data sees no change.  Held here: the 9-layer chain at H = 256 against fp64 with
      small_activations  default-init weights (order 1/16), biases scaled so that every layer's activations are of order 1e-3
      tiny_activations   the same at order 1e-4
      small_weights      weights x 0.05 (order 3e-3), O(1) biases and inputs
      mixed_weights      a different factor per layer (0.02 .. 300): every layer gets its own power of two
    at the same 1e-5 (f16x3) / 2e-6 (f32) as the O(1) cases.  (An unscaled split: 2e-5 and 2e-4 on the first two.)"""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    H, n_msg, n_crd, n_nodes, deg, n_in, D = 256, 4, 5, 400, 12, 24, 6
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(78)
    lin0 = torch.nn.Linear(2 * n_in + 1, H)
    msg = [torch.nn.Linear(H, H) for _ in range(n_msg)]
    crd = [torch.nn.Linear(H, H) for _ in range(n_crd)]
    out = torch.nn.Linear(H, 1, bias=False)
    w_scale, b_scale, in_scale = {"small_activations": (1.0, 0.03, 0.01), "tiny_activations": (1.0, 0.003, 0.001),
                                  "small_weights": (0.05, 1.0, 1.0), "mixed_weights": (None, 1.0, 1.0)}[case]
    mixed = [0.05, 3.0, 0.3, 300.0, 0.02, 2.0, 0.004, 40.0, 1.0]          # (gain 0.3 per unit: the product stays tame)
    with torch.no_grad():
        lin0.weight.mul_(in_scale)
        lin0.bias.mul_(in_scale)
        for k, layer in enumerate(msg + crd):
            layer.weight.mul_(mixed[k] if w_scale is None else w_scale)
            layer.bias.mul_(b_scale)
    degree = torch.randint(1, 2 * deg, (n_nodes,), generator=g)
    src = torch.repeat_interleave(torch.arange(n_nodes), degree)
    E = int(src.numel())
    edges = torch.stack([src, torch.randint(0, n_nodes, (E,), generator=g)], 1)
    h = torch.randn(n_nodes, n_in, generator=g)
    coord = torch.rand(n_nodes, D, generator=g) * 2 - 1
    want_m, want_s = _chain_reference(lin0, msg, crd, out, n_in, h, coord, edges)
    if case.endswith("activations"):      # the regime the case is about
        assert float(want_m.abs().mean()) < (3e-3 if case == "small_activations" else 3e-4)
    mods = [m.to(cuda) for m in [lin0] + msg + crd + [out]]
    pack = kernels.EdgeChainPack(mods[0], mods[1:1 + n_msg], mods[1 + n_msg:-1], mods[-1], input_size=n_in, precision=precision)
    w = mods[0].weight.detach()
    proj = torch.nn.functional.linear(h.to(cuda), torch.cat([w[:, :n_in], w[:, n_in:2 * n_in]], 0)).contiguous()
    status = torch.zeros(1, dtype=torch.int32, device=cuda)
    got_m, got_s = kernels.egnn_edge_chain(pack, proj, coord.to(cuda).contiguous(), edges.to(cuda), status=status)
    pieces, got_s2 = kernels.egnn_edge_chain(pack, proj, coord.to(cuda).contiguous(), edges.to(cuda), status=status, piece_sums=True)
    offsets = (torch.cumsum(degree, 0) - degree).to(cuda)
    got_sum = kernels.segment_combine(pieces, E, offsets, degree.to(cuda), False)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    want_sum = torch.zeros(n_nodes, H, dtype=torch.float64).index_add_(0, src, want_m.detach())
    tol = TOLERANCE[precision]
    errs = (_rel_l2(got_m, want_m), _rel_l2(got_s, want_s), _rel_l2(got_sum, want_sum))
    print(f"{case} / {precision}: messages {errs[0]:.2e}, head {errs[1]:.2e}, node sums {errs[2]:.2e}")
    assert max(errs) < tol, (case, precision, errs)
    assert torch.equal(got_s, got_s2)


def test_edge_chain_f16_range_is_reported(cuda):
    """Split-f16 mode: an activation beyond the f16 range sets MDX_STATUS_EGNN_F16_RANGE; binary32 mode does not care."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    H, n_in = 32, 4
    torch.manual_seed(0)
    lin0, msg, crd, out = torch.nn.Linear(2 * n_in + 1, H), [torch.nn.Linear(H, H)], [torch.nn.Linear(H, H)], \
        torch.nn.Linear(H, 1, bias=False)
    with torch.no_grad():
        lin0.bias.fill_(1e5)                            # SiLU(1e5) = 1e5 > 65504
    mods = [m.to(cuda) for m in (lin0, msg[0], crd[0], out)]
    edges = torch.tensor([[0, 1], [1, 0]], device=cuda)
    proj = torch.zeros(2, 2 * H, device=cuda)
    coord = torch.rand(2, 6, device=cuda)
    for precision, want in (("f16x3", _hip.STATUS_EGNN_F16_RANGE), ("f32", 0)):
        pack = kernels.EdgeChainPack(mods[0], [mods[1]], [mods[2]], mods[3], input_size=n_in, precision=precision)
        status = torch.zeros(1, dtype=torch.int32, device=cuda)
        kernels.egnn_edge_chain(pack, proj, coord, edges, status=status)
        assert int(status.item()) == want


@pytest.mark.parametrize("H", [64, 256])
def test_weight_stream_addresses_verified(cuda, H):
    """Regression guard for the round-2 device fault (a weight-stream request issued with a stale scalar base read unmapped
    memory): csrc/libmdx_hip_verify.so is the same library with the edge chain compiled -DMDX_CHAIN_VERIFY -- every request
    of every wavefront compares the global and LDS addresses it is about to use with the plain formula and reports a
    mismatch in bits 30 / 31 of the status word.  The split-f16 chain (rows and piece-sums instantiations, several tiles per
    workgroup, a ragged tail) runs once through it: no bit set, and the same outputs, bit for bit, as the shipped library."""
    import os
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    verify_path = os.path.join(os.path.dirname(_hip.LIB_PATH), "libmdx_hip_verify.so")
    assert os.path.exists(verify_path), "build it with `make -C csrc verify` (part of `make all` and of __graft_entry__.build())"
    shipped_path, shipped = _hip.LIB_PATH, _hip.lib()
    n_msg, n_crd, n_in, D, n_nodes = 2, 3, 16, 6, 3000
    g = torch.Generator().manual_seed(H)
    torch.manual_seed(H + 1)
    mods = [m.to(cuda) for m in [torch.nn.Linear(2 * n_in + 1, H)] + [torch.nn.Linear(H, H) for _ in range(n_msg + n_crd)] +
            [torch.nn.Linear(H, 1, bias=False)]]
    degree = torch.randint(0, 40, (n_nodes,), generator=g)
    src = torch.repeat_interleave(torch.arange(n_nodes), degree)
    edges = torch.stack([src, torch.randint(0, n_nodes, (src.numel(),), generator=g)], 1).to(cuda)
    proj = torch.randn(n_nodes, 2 * H, generator=g).to(cuda)
    coord = torch.rand(n_nodes, D, generator=g).to(cuda)
    outs = {}
    try:
        for name, path in (("shipped", shipped_path), ("verify", verify_path)):
            _hip._lib, _hip.LIB_PATH = (shipped, path) if name == "shipped" else (None, path)
            _hip.lib()
            pack = kernels.EdgeChainPack(mods[0], mods[1:1 + n_msg], mods[1 + n_msg:-1], mods[-1], input_size=n_in, precision="f16x3")
            status = torch.zeros(1, dtype=torch.int32, device=cuda)
            rows = kernels.egnn_edge_chain(pack, proj, coord, edges, status=status, piece_sums=False)
            pieces = kernels.egnn_edge_chain(pack, proj, coord, edges, status=status, piece_sums=True)
            torch.cuda.synchronize()
            word = int(status.item()) & 0xFFFFFFFF
            assert word == 0, f"{name}: status {word:#x} (bit 30: source address, bit 31: LDS address of a request)"
            outs[name] = (rows[0], rows[1], pieces[1])
    finally:
        _hip._lib, _hip.LIB_PATH = shipped, shipped_path
    for a, b in zip(outs["shipped"], outs["verify"]):
        assert torch.equal(a, b)


def test_coord_aggregate_against_torch(cuda):
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    g = torch.Generator().manual_seed(5)
    n_nodes, D = 300, 6
    degree = torch.randint(0, 30, (n_nodes,), generator=g)
    degree[7] = 0
    src = torch.repeat_interleave(torch.arange(n_nodes), degree)
    E = int(src.numel())
    dst = torch.randint(0, n_nodes, (E,), generator=g)
    s = torch.randn(E, generator=g)
    coord = torch.randn(n_nodes, D, generator=g)
    offsets = torch.cumsum(degree, 0) - degree
    for mean in (False, True):
        want = torch.zeros(n_nodes, D, dtype=torch.float64).index_add_(
            0, src, (coord.double()[src] - coord.double()[dst]) * s.double()[:, None])
        if mean:
            want = want / degree.clamp(min=1).double()[:, None]
        want = coord.double() + want
        got = kernels.egnn_coord_aggregate(s.to(cuda), coord.to(cuda), torch.stack([src, dst], 1).to(cuda),
                                           offsets.to(cuda), degree.to(cuda), mean)
        assert _rel_l2(got, want) < 1e-6


@pytest.mark.parametrize("hidden,n_layers,n_hidden,B,N,box", [(32, 2, 2, 6, 64, 11.084), (128, 3, 3, 6, 64, 11.084),
                                                              (256, 4, 4, 6, 64, 11.084), (64, 2, 2, 2, 500, 22.0)])
def test_egnn_forward_edge_chain_modes(cuda, hidden, n_layers, n_hidden, B, N, box):
    """EGNNScoreNetwork.forward (radius graph, N = 64 and one 500-atom case, two atom types) with the fused MFMA edge chain -- exact binary32
    and split-f16 -- against the per-layer library-GEMM path and against the plain PyTorch module: scores and logits
    within 1e-5 of the output scale; fp64 evaluation of the same module as the common yardstick."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    torch.manual_seed(7)
    net = nets.egnn_net(2, "radial_cutoff", 7.5, hidden=hidden, n_layers=n_layers, n_hidden=n_hidden).to(cuda)
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, 3, (B, N), device=cuda), X=torch.rand(B, N, 3, device=cuda),
                                        L=torch.tensor([box] * 3 + [0.0] * 3, device=cuda).repeat(B, 1)),
             TIME: torch.rand(B, 1, device=cuda), NOISE: torch.rand(B, 1, device=cuda) * 0.2,
             CARTESIAN_FORCES: torch.zeros(B, N, 3, device=cuda)}
    outs = {}
    for mode in ("f32", "f16x3", "f16x3_32x32", None, "plain"):
        net.edge_chain_precision = None if mode == "plain" else mode
        for layer in net.egnn.graph_layers:
            layer.use_fused_ops = mode != "plain"
        with torch.no_grad():
            outs[mode] = net(batch, conditional=False)
        net.check_status()
    assert net.egnn.graph_layers[0]._chain[1] is not None           # the MFMA kernel really ran
    layer0 = net.egnn.graph_layers[0]                               # ... and the node MLP on the same pipeline
    assert layer0._node_mlp[1] is not None or layer0._node_chain[1] is not None
    # fp64 yardstick: the same module in double precision on the same edges
    import copy
    net64 = copy.deepcopy(net).double()
    for layer in net64.egnn.graph_layers:
        layer.use_fused_ops = False
    batch64 = {k: (AXL(A=v.A, X=v.X.double(), L=v.L.double()) if k == NOISY_AXL_COMPOSITION else v.double())
               for k, v in batch.items()}
    edges, degree = net._build_edges(batch[NOISY_AXL_COMPOSITION].X, batch[NOISY_AXL_COMPOSITION].L)
    net64.edge_builder = lambda x, cell, rc: (edges, degree)
    with torch.no_grad():
        want = net64(batch64, conditional=False)
    errs = {}
    for mode, got in outs.items():
        errs[mode] = (_rel_l2(got.X, want.X), _rel_l2(got.A[..., :-1], want.A[..., :-1]))
    print("EGNN forward rel-L2 vs fp64 (scores, logits):", {str(k): tuple(f"{e:.2e}" for e in v) for k, v in errs.items()})
    # (1) against the REFERENCE ARITHMETIC -- the plain PyTorch fp32 module on the same device: north_star's 1e-5.  (The
    # reference-made fixture at the production shape is tests/test_egnn_c3_reference_gpu.py; this test sweeps widths.)
    # In the 22 A cell of the 500-atom case binary32 itself is the limit: the plain module sits 3.3e-5 from its own fp64
    # evaluation (the update x + trans rounds at the magnitude of x; DESIGN.md section 5, the configs[4] forward), and two
    # binary32 evaluations in different orders cannot be held closer to each other than that floor.
    plain = outs["plain"]
    tolerance = max(1e-5, errs["plain"][0])
    for mode in ("f32", "f16x3", "f16x3_32x32", None):
        vs_plain = (_rel_l2(outs[mode].X, plain.X), _rel_l2(outs[mode].A[..., :-1], plain.A[..., :-1]))
        assert vs_plain[0] < tolerance and vs_plain[1] < 1e-5, (mode, vs_plain)
    # (2) against fp64: every fp32 evaluation of this network -- the reference's included (1.4e-5 at the production shape,
    # DESIGN.md section 3a) -- carries the same rounding of the coordinate update `x + trans` (x of order one, the score is
    # extracted from the small update), 3e-6 .. 9e-6 here.  The fused paths must not be further from fp64 than plain fp32 is.
    for mode in ("f32", "f16x3", "f16x3_32x32", None):
        assert errs[mode][0] < 1.25 * errs["plain"][0] + 1e-7 and errs[mode][1] < 1.25 * errs["plain"][1] + 1e-7, (mode, errs)
    assert errs["plain"][0] < (1.5e-5 if box < 12 else 5e-5) and errs["plain"][1] < 1e-5, errs


@pytest.mark.parametrize("rc,expect_edges,N", [(3.0, True, 16), (0.5, False, 16), (3.0, False, 1)])
@pytest.mark.parametrize("two_call", [False, True])
def test_egnn_forward_sparse_and_empty_graphs(cuda, rc, expect_edges, two_call, N):
    """Dilute structures (16 atoms in a 30 A cell): most atoms have no neighbour inside the cutoff, whole structures have no
    edge, and with rc = 0.5 the batch's graph is EMPTY.  The reference's segment sums then add nothing and every layer reduces
    to the node MLP on [h | 0]; the fused path (radius graph with zero counts, edge chain on zero tiles, piece sums of empty
    segments) must return what the plain PyTorch module returns, in every arithmetic mode, with the capacity-sized edge list
    and with the two-call radius graph."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    torch.manual_seed(11)
    net = nets.egnn_net(2, "radial_cutoff", rc, hidden=64, n_layers=2, n_hidden=2).to(cuda)
    if two_call:
        net.static_edge_list_max_fraction = 0.0
    B = 5                                                                     # (N = 1: a structure of a single atom)
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, 3, (B, N), device=cuda), X=torch.rand(B, N, 3, device=cuda),
                                        L=torch.tensor([30.0] * 3 + [0.0] * 3, device=cuda).repeat(B, 1)),
             TIME: torch.rand(B, 1, device=cuda), NOISE: torch.rand(B, 1, device=cuda) * 0.2,
             CARTESIAN_FORCES: torch.zeros(B, N, 3, device=cuda)}
    x = batch[NOISY_AXL_COMPOSITION].X.double()
    delta = x[:, :, None, :] - x[:, None, :, :]
    r = ((delta - delta.round()) * 30.0).norm(dim=-1)
    degree = ((r <= rc) & (r > 0)).sum(-1)                                    # brute-force minimum-image degrees
    assert (int(degree.sum()) > 0) == expect_edges and int((degree == 0).sum()) >= B * N // 2
    outs = {}
    for mode in ("f32", "f16x3", None, "plain"):
        net.edge_chain_precision = None if mode == "plain" else mode
        for layer in net.egnn.graph_layers:
            layer.use_fused_ops = mode != "plain"
        with torch.no_grad():
            outs[mode] = net(batch, conditional=False)
        net.check_status()
        assert torch.isfinite(outs[mode].X).all() and torch.isfinite(outs[mode].A[..., :-1]).all(), mode
    plain = outs["plain"]
    if not expect_edges:                         # no edge, no coordinate update: the score is an exact zero on both paths
        assert (plain.X == 0).all()
    for mode in ("f32", "f16x3", None):
        assert _rel_l2(outs[mode].A[..., :-1], plain.A[..., :-1]) < 1e-5, mode
        if expect_edges:
            assert _rel_l2(outs[mode].X, plain.X) < 1e-5, mode
        else:
            assert (outs[mode].X == 0).all(), mode


def test_egnn_conditional_forward_keeps_the_mask_logit(cuda):
    """ScoreNetwork.forward(conditional=True) blends two evaluations (score_network.py:187-223); on the fused path each of them
    already carries the MASK logit at -inf (mdx_egnn_outputs), and -inf times a zero weight is a NaN: the base class's
    assignment must still run on the blend.  gamma = 1 is that case."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    torch.manual_seed(3)
    net = nets.egnn_net(2, "radial_cutoff", 3.0, hidden=32, n_layers=2, n_hidden=2).to(cuda)
    B, N = 5, 16
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, 3, (B, N), device=cuda), X=torch.rand(B, N, 3, device=cuda),
                                        L=torch.tensor([7.0, 7.0, 7.0, 0, 0, 0], device=cuda).repeat(B, 1)),
             TIME: torch.rand(B, 1, device=cuda), NOISE: torch.rand(B, 1, device=cuda) * 0.3 + 0.01,
             CARTESIAN_FORCES: torch.zeros(B, N, 3, device=cuda)}
    with torch.no_grad():
        plain = net(batch, conditional=False)
        for gamma in (1.0, 0.0, 0.4):
            net.conditional_gamma = gamma
            out = net(batch, conditional=True)
            assert torch.isinf(out.A[..., -1]).all() and (out.A[..., -1] < 0).all(), gamma
            assert torch.isfinite(out.A[..., :-1]).all() and torch.isfinite(out.X).all(), gamma
            assert torch.allclose(out.X, plain.X, rtol=1e-6, atol=1e-9) and torch.allclose(out.A[..., :-1], plain.A[..., :-1], rtol=1e-6)
    net.check_status()


def test_egnn_fused_path_is_off_under_autograd(cuda):
    """With gradients enabled the module runs as plain PyTorch (the HIP calls are invisible to autograd): outputs carry a
    grad_fn and match the no-grad fused result."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    torch.manual_seed(3)
    net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=1).to(cuda)
    B, N = 2, 64
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, 2, (B, N), device=cuda), X=torch.rand(B, N, 3, device=cuda),
                                        L=torch.tensor([10.86] * 3 + [0.0] * 3, device=cuda).repeat(B, 1)),
             TIME: torch.rand(B, 1, device=cuda), NOISE: torch.rand(B, 1, device=cuda) * 0.2,
             CARTESIAN_FORCES: torch.zeros(B, N, 3, device=cuda)}
    with torch.no_grad():
        fused = net(batch, conditional=False)
    with_grad = net(batch, conditional=False)
    assert with_grad.X.grad_fn is not None and fused.X.grad_fn is None
    assert _rel_l2(with_grad.X.detach(), fused.X) < 1e-4          # same function, two fp32 evaluation orders
    with_grad.X.sum().backward()
    assert net.egnn.graph_layers[0].message_mlp[2].weight.grad is not None


def test_egnn_accepts_unsorted_edges(cuda):
    """A caller's own edge list in any order (the reference's unsorted_segment_sum accepts it): EGNN.forward sorts it by
    source before the segment kernels; result equals the sorted call."""
    torch.manual_seed(5)
    net = nets.egnn_net(1, "fully_connected", None, hidden=32, n_layers=2, n_hidden=1).to(cuda)
    n_nodes = 24
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils import neighbors
    edges = neighbors.get_edges_batch(8, 3, device=cuda)
    h = torch.randn(n_nodes, 3, device=cuda)
    x = torch.randn(n_nodes, 6, device=cuda)
    perm = torch.randperm(edges.shape[0], device=cuda)
    with torch.no_grad():
        a = net.egnn(h=h, edges=edges, x=x.clone())
        b = net.egnn(h=h, edges=edges[perm], x=x.clone())
    assert _rel_l2(b.X, a.X) < 1e-5 and _rel_l2(b.A, a.A) < 1e-5


def test_radius_graph_static_equals_two_call(cuda):
    """The capacity-sized radius graph (no host read between count and fill) gives the same edge list, degrees and count as
    the two-call protocol; a capacity that is too small writes nothing beyond it and reports STATUS_GRAPH_CAPACITY."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    torch.manual_seed(11)
    B, N, rc = 7, 64, 7.5
    x = torch.rand(B, N, 3, device=cuda)
    cell = torch.diag(torch.tensor([16.5, 16.5, 16.5])).repeat(B, 1, 1).to(cuda)
    cart = (x @ cell).contiguous()
    ref = kernels.radius_graph(cart, cell, rc, unique=True)
    E = ref["edges"].shape[0]
    status = torch.zeros(1, dtype=torch.int32, device=cuda)
    out = kernels.radius_graph_static(cart, cell, rc, B * N * (N - 1), status=status)
    assert int(out["n_edges"].item()) == E and int(status.item()) == 0
    assert torch.equal(out["edges"][:E], ref["edges"]) and torch.equal(out["counts"], ref["counts"].view(-1))
    assert torch.equal(out["offsets"], torch.cumsum(ref["counts"].view(-1), 0) - ref["counts"].view(-1))
    small = E - 37
    guard = torch.full((small + 64, 2), -5, dtype=torch.int64, device=cuda)
    check, lib, ptr, stream_handle = _hip.check, _hip.lib, _hip.ptr, _hip.stream_handle
    check(lib().mdx_radius_graph_fill_capped(ptr(cart, torch.float32, "c"), ptr(cell, torch.float32, "b"), rc, B, N, 1,
                                             ptr(out["offsets"], torch.int64, "o"), small, ptr(guard, torch.int64, "e"), None,
                                             None, ptr(status, torch.int32, "s"), stream_handle()), "fill_capped")
    assert int(status.item()) == _hip.STATUS_GRAPH_CAPACITY
    assert torch.equal(guard[:small], ref["edges"][:small]) and (guard[small:] == -5).all()


@pytest.mark.parametrize("two_launches", [True, False], ids=["masks_emit", "count_scan_fill"])
@pytest.mark.parametrize("B,N", [(3, 5), (7, 64), (70, 64), (2, 200), (257, 65), (520, 64), (2, 513), (1, 1000), (1, 1), (5, 63),
                                 (3, 128), (256, 216), (2, 1024), (2, 1025), (2049, 8), (2048, 8)])
def test_egnn_radius_graph_from_relative_coordinates(cuda, B, N, two_launches):
    """mdx_egnn_radius_graph (relative coordinates + lattice parameters in, clip / diagonal cell / positions / scan inside) gives
    bit for bit what the score network built before from torch.clip, diag_embed, matmul, the two radius-graph launches and
    torch.cumsum, in both of its forms -- hit masks + emission (two launches with a workspace; a source row of 1 - 16 words, one or
    sixteen wavefronts per structure, the shapes C3 and C5 sample) and count / scan / fill (three; also what the first form falls
    back to beyond N 1024 or B 2048) -- lengths below the clip, count lists of more than one scan tile of 16 384 entries (257 x 65
    with a ragged last thread, 520 x 64 with three tiles) and short ragged ones included; a capacity that is too small is reported
    and respected."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    g = torch.Generator().manual_seed(100 * B + N)
    rc = 3.2
    x = torch.rand(B, N, 3, generator=g).to(cuda)
    lattice = torch.cat([torch.rand(B, 3, generator=g) * 9.0 + 4.0, torch.zeros(B, 3)], dim=1).to(cuda)   # some < 2.2 rc = 7.04
    lengths = lattice[:, :3].clip(min=2.2 * rc)
    cell = torch.diag_embed(lengths)
    cart = torch.matmul(x, cell).contiguous()
    status = torch.zeros(1, dtype=torch.int32, device=cuda)
    capacity = B * N * (N - 1)
    want = kernels.radius_graph_static(cart, cell.contiguous(), rc, capacity, status=status)
    words = int(_hip.lib().mdx_egnn_radius_graph_workspace_words(B, N))
    assert (words == B * N * ((N + 63) // 64) + B) if (N <= 1024 and B <= 2048) else words == 0
    got = kernels.egnn_radius_graph(x, lattice, 2.2 * rc, rc, capacity, status=status, two_launches=two_launches)
    E = int(want["n_edges"].item())
    assert int(got["n_edges"].item()) == E and int(status.item()) == 0 and (E > 0 or N == 1)
    assert torch.equal(got["counts"], want["counts"]) and torch.equal(got["offsets"], want["offsets"])
    assert torch.equal(got["edges"][:E], want["edges"][:E])
    if E > 3:
        guard = kernels.egnn_radius_graph(x, lattice, 2.2 * rc, rc, E - 3, status=status, two_launches=two_launches)
        assert int(status.item()) == _hip.STATUS_GRAPH_CAPACITY and torch.equal(guard["edges"], want["edges"][:E - 3])
        assert int(guard["n_edges"].item()) == E          # the count is the graph's, not the list's


def test_egnn_radius_graph_forms_on_random_shapes_and_small_cells(cuda):
    """The two forms of mdx_egnn_radius_graph against each other and against the two-call search over 40 random (B, N, cutoff,
    clip) -- including clips BELOW 2.2 x cutoff, where the cell no longer guarantees a single image: the mask kernel then sweeps
    the 27 images like radius_graph_kernel (a pair within the cutoff through several images is still ONE edge), and a cell shorter
    than the cutoff raises the same status bit in both forms."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    g = torch.Generator().manual_seed(2718)
    swept = too_large = 0
    for case in range(40):
        B = int(torch.randint(1, 40, (1,), generator=g))
        N = int(torch.randint(1, 300, (1,), generator=g))
        rc = float(torch.rand(1, generator=g)) * 3.0 + 1.5
        clip = rc * (2.2 if case % 2 == 0 else float(torch.rand(1, generator=g)) * 1.6 + 0.7)        # 0.7 .. 2.3 x cutoff
        x = torch.rand(B, N, 3, generator=g)
        if case % 5 == 3:           # coordinates outside [0, 1), coincident atoms, a NaN and an infinity: the same edges either way
            x = x * 3.0 - 1.0
            x[0, -1] = x[0, 0]
            if N > 3:
                x[-1, 1, 0], x[-1, 2, 2] = float("nan"), float("inf")
        x = x.to(cuda)
        lattice = torch.cat([torch.rand(B, 3, generator=g) * 8.0 + 2.0, torch.zeros(B, 3)], dim=1).to(cuda)
        cell = torch.diag_embed(lattice[:, :3].clip(min=clip)).contiguous()
        cart = torch.matmul(x, cell).contiguous()
        capacity = B * N * (N - 1)
        st = [torch.zeros(1, dtype=torch.int32, device=cuda) for _ in range(3)]
        want = kernels.radius_graph_static(cart, cell, rc, capacity, status=st[0])
        two = kernels.egnn_radius_graph(x, lattice, clip, rc, capacity, status=st[1], two_launches=True)
        three = kernels.egnn_radius_graph(x, lattice, clip, rc, capacity, status=st[2], two_launches=False)
        assert int(st[0].item()) == int(st[1].item()) == int(st[2].item()), (case, B, N, rc, clip)
        E = int(want["n_edges"].item())
        for got in (two, three):
            assert int(got["n_edges"].item()) == E
            assert torch.equal(got["counts"], want["counts"]) and torch.equal(got["offsets"], want["offsets"])
            assert torch.equal(got["edges"][:E], want["edges"][:E]), (case, B, N, rc, clip)
        swept += int(bool((cell.diagonal(dim1=1, dim2=2).min() < 2.2 * rc).item()))
        too_large += int(int(st[1].item()) == _hip.STATUS_CUTOFF_TOO_LARGE)
    assert swept >= 10 and too_large >= 2          # the sweep path and the status were exercised


@pytest.mark.parametrize("H,C,n_nodes", [(256, 2, 1000), (64, 3, 77), (32, 8, 5)])
def test_egnn_outputs_against_torch(cuda, H, C, n_nodes):
    """mdx_egnn_outputs: the classification layer with the MASK logit at -inf, the scores (the bits of mdx_egnn_scores) and the
    zero lattice output in one launch."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import \
        positive_bloch_wave_vectors
    g = torch.Generator().manual_seed(H + C)
    kv = positive_bloch_wave_vectors(1, 3).to(cuda)
    z = torch.randn(n_nodes, 2 * kv.shape[0], generator=g).to(cuda)
    x_hat = torch.randn(n_nodes, 2 * kv.shape[0], generator=g).to(cuda)
    h = torch.randn(n_nodes, H, generator=g).to(cuda)
    head = torch.nn.Linear(H, C).to(cuda)
    scores, logits, zeros = kernels.egnn_outputs(z, x_hat, kv, h, head.weight.detach(), head.bias.detach(), C - 1, 42)
    assert torch.equal(scores, kernels.egnn_scores(z, x_hat, kv))
    want = torch.nn.functional.linear(h.double(), head.weight.detach().double(), head.bias.detach().double())
    assert _rel_l2(logits[:, :C - 1], want[:, :C - 1]) < 1e-6
    assert torch.isinf(logits[:, C - 1]).all() and (logits[:, C - 1] < 0).all()
    assert zeros.shape == (42,) and not zeros.any()
    _, plain, _ = kernels.egnn_outputs(z, x_hat, kv, h, head.weight.detach(), head.bias.detach(), -1, 0)
    assert _rel_l2(plain, want) < 1e-6


@pytest.mark.parametrize("precision", CHAIN_MODES)
def test_egnn_sampler_graph_replay_equals_eager(cuda, precision):
    """The EGNN sampler iteration (radius graph with a capacity-sized edge list + fused edge chain: no host read) captured
    into a hipGraph and replayed equals the eager run bit for bit; and equals the run with the two-call radius graph."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    import cases
    import warnings
    outs = {}
    for mode in ("eager", "graph", "two_call"):
        torch.manual_seed(21)
        net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=2).to(cuda)
        net.edge_chain_precision = precision
        if mode == "two_call":
            net.static_edge_list_max_fraction = 0.0      # force the two-call protocol
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = NoiseParameters(**cases.noise_ns(5, **cases.LIN))
            spar = PredictorCorrectorSamplingParameters(**cases.sampling_ns(64, 1, M=2, greedy=False, one=False,
                                                                            cell=[10.86] * 3),
                                                        rng_mode="device", seed=31, use_hip_graph=mode == "graph")
        gen = LangevinGenerator(npar, spar, net)
        with torch.no_grad():
            out = gen.sample(12, cuda)
        outs[mode] = (out.A.cpu().numpy(), out.X.cpu().numpy())
    for mode in ("graph", "two_call"):
        assert np.array_equal(outs["eager"][0], outs[mode][0]), mode
        assert np.array_equal(outs["eager"][1].view(np.int32), outs[mode][1].view(np.int32)), mode
    assert (outs["eager"][0] != 1).all()


@pytest.mark.parametrize("rng_mode", ["device", "reference"])
def test_sampler_recomputes_in_f32_when_the_f16_range_is_left(cuda, rng_mode):
    """A network whose activations leave the f16 range at EVERY time index (first-layer bias 7e4): the split-f16 kernels report
    it and the FIRST iteration is recomputed with the exact-f32 kernels on the same draws (device RNG: a draw is a function of
    the index; reference-order RNG: the iteration's draws are kept and handed out again), with a warning and a count; that f32
    pass records the activation maxima, the exponents of the hot positions are lowered, and the remaining iterations run the
    split-f16 kernels -- the result is within 1e-5 of a run that used 'f32' from the start, atom types equal.  The switch is
    local: the network's setting is 'f16x3' again afterwards, a recorded trajectory holds the kept steps only, and the next
    call -- whose activations stay in range -- runs the split-f16 kernels with no retry."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    import cases
    import warnings
    T, B = 3, 6
    outs, gens = {}, {}
    for mode in ("f16x3", "f32"):
        torch.manual_seed(21)
        net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=2).to(cuda)
        first_bias = net.egnn.graph_layers[0].message_mlp[0].bias
        normal_bias = first_bias.detach().clone()
        with torch.no_grad():
            first_bias.fill_(7.0e4)                                            # SiLU(7e4) = 7e4: beyond the carried f16 range
            for lin in net.egnn.graph_layers[0].message_mlp[2::2]:
                lin.weight.mul_(1e-4)                                          # keep everything downstream finite in f32
        net.edge_chain_precision = mode
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = NoiseParameters(**cases.noise_ns(T, **cases.LIN))
            spar = PredictorCorrectorSamplingParameters(**cases.sampling_ns(64, 1, M=1, greedy=False, one=False,
                                                                            cell=[10.86] * 3), rng_mode=rng_mode, seed=5,
                                                        record_samples=True)
        gen = gens[mode] = LangevinGenerator(npar, spar, net)
        torch.manual_seed(99)                                                  # (the reference-order draws)
        with torch.no_grad():
            if mode == "f16x3":
                with pytest.warns(UserWarning, match="f16 range"):
                    first = gen.sample(B, cuda)
                assert net.edge_chain_precision == "f16x3" and gen.f16_range_fallbacks == 1     # (one ITERATION, then adapted)
            else:
                first = gen.sample(B, cuda)
                assert gen.f16_range_fallbacks == 0
            # the discarded attempt left nothing in the recorder: T predictor steps, not 2 T
            assert len(gen.sample_trajectory_recorder._internal_data["predictor_step"]) == T
            # second call, activations in range: no retry, split-f16 kernels
            first_bias.copy_(normal_bias)
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                second = gen.sample(B, cuda)
        outs[mode] = (first, second)
        assert gen.f16_range_fallbacks == (1 if mode == "f16x3" else 0) and net.edge_chain_precision == mode
        assert all(layer._chain[1].precision == mode for layer in net.egnn.graph_layers)      # what the last forward ran
    assert torch.isfinite(outs["f32"][0].X).all()
    # both calls ran (mostly) different arithmetic on the same draws: equal atom types, coordinates within the tolerance
    for call in (0, 1):
        assert torch.equal(outs["f16x3"][call].A, outs["f32"][call].A)
        diff = (outs["f16x3"][call].X - outs["f32"][call].X + 0.5) % 1.0 - 0.5
        assert float(diff.norm() / outs["f32"][call].X.norm()) < 1e-5, call


def test_activation_exponents_adapt_to_a_hot_layer(cuda):
    """Per-position activation exponents (mdx_egnn_chain_t.activation_exponents).  A chain whose first
    layer's output is ~3 000 (carried value ~4 300: beyond 65504 / 2^6 = 1023): the split-f16 kernel reports the range bit with
    the default exponents; the exact-f32 kernel, sharing the chain's ActivationScales, records the largest carried value per
    position; mdx_egnn_chain_adapt_activation_exponents lowers the exponent of the hot position (2^e max in [2^12, 2^13)) and
    leaves the others at 6; the split-f16 kernel then runs clean and agrees with fp64 to the usual tolerance -- one hot layer
    no longer sends the chain to the f32 kernels."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    H, n_in, n_nodes = 64, 8, 300
    torch.manual_seed(5)
    lin0 = torch.nn.Linear(2 * n_in + 1, H)
    msg, crd = [torch.nn.Linear(H, H) for _ in range(2)], [torch.nn.Linear(H, H) for _ in range(2)]
    out = torch.nn.Linear(H, 1, bias=False)
    with torch.no_grad():
        lin0.bias.add_(3000.0)                  # SiLU(~3000) ~ 3000 at position 0
        msg[0].weight.mul_(1.0e-3)              # the next layer brings the values back to order one
    mods = [m.to(cuda) for m in [lin0] + msg + crd + [out]]
    g = torch.Generator().manual_seed(6)
    degree = torch.randint(1, 30, (n_nodes,), generator=g)
    src = torch.repeat_interleave(torch.arange(n_nodes), degree)
    edges = torch.stack([src, torch.randint(0, n_nodes, (src.numel(),), generator=g)], 1).to(cuda)
    h = torch.randn(n_nodes, n_in, generator=g)
    coord = torch.rand(n_nodes, 6, generator=g)
    scales = kernels.ActivationScales(4, cuda)
    packs = {prec: kernels.EdgeChainPack(mods[0], mods[1:3], mods[3:5], mods[5], input_size=n_in, precision=prec, scales=scales)
             for prec in ("f16x3", "f32")}
    proj = torch.nn.functional.linear(h.to(cuda), packs["f32"].proj_weight).contiguous()

    def run(prec):
        status = torch.zeros(1, dtype=torch.int32, device=cuda)
        m, s = kernels.egnn_edge_chain(packs[prec], proj, coord.to(cuda), edges, status=status)
        return m, s, int(status.item())

    assert run("f16x3")[2] == _hip.STATUS_EGNN_F16_RANGE
    assert (scales.exponents == 6).all() and (scales.maxima == 0).all()
    m32, s32, st = run("f32")
    assert st == 0
    maxima = scales.maxima.view(torch.float32).cpu()
    assert 3000 < float(maxima[0]) < 6000 and (maxima[1:5] > 0).all() and (maxima[1:5] < 100).all() and maxima[5] == 0
    scales.adapt()
    exps = scales.exponents.cpu().tolist()
    assert exps[0] == 12 - int(np.floor(np.log2(float(maxima[0])))) == 0 and exps[1:] == [6] * 5
    assert (scales.maxima == 0).all()
    m16, s16, st = run("f16x3")
    assert st == 0, "the adapted exponents keep the hot position inside the f16 range"
    # against fp64
    d = lambda t: t.detach().double().cpu()       # noqa: E731
    x = torch.nn.functional.silu(d(proj)[src, :H] + d(proj)[edges[:, 1].cpu(), H:] + d(mods[0].bias) +
                                 ((coord[src] - coord[edges[:, 1].cpu()]).double() ** 2).sum(1, keepdim=True) * d(mods[0].weight)[:, 2 * n_in])
    for lin in mods[1:3]:
        x = torch.nn.functional.silu(x @ d(lin.weight).T + d(lin.bias))
    want_m, y = x, x
    for lin in mods[3:5]:
        y = torch.nn.functional.silu(y @ d(lin.weight).T + d(lin.bias))
    want_s = (y @ d(mods[5].weight).T).reshape(-1)
    for got_m, got_s, tol in ((m16, s16, 1e-5), (m32, s32, 2e-6)):
        assert _rel_l2(got_m, want_m) < tol and _rel_l2(got_s, want_s) < tol
    # an exponent only goes down: a later, cooler f32 pass leaves it where it is
    with torch.no_grad():
        mods[0].bias.sub_(3000.0)
    packs = {prec: kernels.EdgeChainPack(mods[0], mods[1:3], mods[3:5], mods[5], input_size=n_in, precision=prec, scales=scales)
             for prec in ("f16x3", "f32")}
    run("f32")
    scales.adapt()
    assert scales.exponents.cpu().tolist() == exps


def test_sampler_adapts_the_activation_exponents_after_one_fallback(cuda):
    """The same through the sampler: an EGNN whose first graph layer runs hot at EVERY time index (first-message-layer bias
    3 000).  Round 3 recomputed the whole call in f32; round 4's first version would recompute every iteration; now the first
    iteration is recomputed once with the f32 kernels, which record the activation maxima, the exponents are adapted on the
    device, and every later iteration -- hipGraph replays included: the kernels read the exponents at launch -- runs the
    split-f16 kernels: ONE fallback over the call, none in the next call, results within 1e-5 of the all-f32 run."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    import cases
    import warnings
    T, B = 6, 5
    outs = {}
    for mode, use_graph in (("f16x3", True), ("f16x3", False), ("f32", False)):
        torch.manual_seed(21)
        net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=2).to(cuda)
        with torch.no_grad():
            net.egnn.graph_layers[0].message_mlp[0].bias.add_(3000.0)
            net.egnn.graph_layers[0].message_mlp[2].weight.mul_(1.0e-3)
        net.edge_chain_precision = mode
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = NoiseParameters(**cases.noise_ns(T, **cases.LIN))
            spar = PredictorCorrectorSamplingParameters(**cases.sampling_ns(64, 1, M=1, greedy=False, one=False, cell=[10.86] * 3),
                                                        rng_mode="device", seed=5, use_hip_graph=use_graph)
        gen = LangevinGenerator(npar, spar, net)
        with torch.no_grad():
            if mode == "f16x3":
                with pytest.warns(UserWarning, match="f16 range"):
                    first = gen.sample(B, cuda)
                assert gen.f16_range_fallbacks == 1, gen.f16_range_fallbacks
                edge = net.egnn.graph_layers[0]._activation_scales
                assert any(int(s.exponents.min()) < 6 for s in edge.values())
                with warnings.catch_warnings():
                    warnings.simplefilter("error")
                    second = gen.sample(B, cuda)
                assert gen.f16_range_fallbacks == 1
            else:
                first, second = gen.sample(B, cuda), gen.sample(B, cuda)
        outs[(mode, use_graph)] = (first, second)
    for key in (("f16x3", True), ("f16x3", False)):
        for got, want in zip(outs[key], outs[("f32", False)]):
            assert torch.equal(got.A, want.A)
            diff = (got.X - want.X + 0.5) % 1.0 - 0.5
            assert float(diff.norm() / want.X.norm()) < 1e-5, key
    # graph replay and eager steps took the same decisions: same bits
    for a, b in zip(outs[("f16x3", True)], outs[("f16x3", False)]):
        assert torch.equal(a.X, b.X)


class _RangeReportAt(torch.nn.Module):
    """Test plugin around an EGNN: at ONE time value it raises the f16-range bit in the network's status word -- what the
    split-f16 kernels do when an activation overflows -- and spoils the scores of that forward, as an overflow would.  Device
    operations only (the iteration is captured into a hipGraph); silent when the network runs the exact-f32 kernels."""

    def __init__(self, net, time_value):
        super().__init__()
        self.net = net
        self.register_buffer("time_values", torch.as_tensor(time_value, dtype=torch.float32).reshape(-1))

    graph_status = property(lambda self: self.net.graph_status)
    edge_chain_precision = property(lambda self: self.net.edge_chain_precision,
                                    lambda self, value: setattr(self.net, "edge_chain_precision", value))

    def forward(self, batch, conditional=None):
        from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
        from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL, TIME
        out = self.net(batch, conditional)
        if self.net.edge_chain_precision in ("f16x3", "f16x3_32x32"):
            hit = (batch[TIME][:1, 0:1] == self.time_values).any().reshape(1)       # (the buffer lives on the device: .to(cuda))
            self.net.graph_status.bitwise_or_(hit.to(torch.int32) * _hip.STATUS_EGNN_F16_RANGE)
            out = AXL(A=out.A, X=out.X + hit.to(out.X.dtype) * 1.0e3, L=out.L)
        return out


@pytest.mark.parametrize("precision,repaint", [("f16x3", False), ("f16x3_32x32", False), ("f16x3", True)])
@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("M,flagged", [(0, 1), (2, 2)])
def test_f16_range_fallback_costs_one_iteration(cuda, M, flagged, use_graph, precision, repaint):
    """The f16-range report is handled per ITERATION: a run whose network reports the range bit
    at one time value only -- seen by ONE iteration with no correctors, by two neighbouring ones with correctors (an
    iteration's correctors and the next iteration's predictor share a time value) -- completes with exactly that many
    iterations recomputed in f32, and its result equals, bit for bit, a run in which exactly those iterations were computed
    with the f32 kernels and all the others with the split-f16 kernels.  Both loops (hipGraph replays watched two iterations
    behind the queue; eager steps), both split modes; the iterations queued behind the flagged one are dropped and repeated.
    `repaint`: the constrained generator with one resampling pass per time index (each iteration = two predictor / corrector
    visits and a forward step; the redone iteration runs all of them eagerly in f32)."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    import cases
    import warnings
    T, B, k = 9, 5, 4                                   # the predictor of iteration k (time index k + 1 -> k) sees time[k]

    def build(wrap):
        torch.manual_seed(21)
        net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=2).to(cuda)
        net.edge_chain_precision = precision
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = NoiseParameters(**cases.noise_ns(T, **cases.LIN))
            spar = PredictorCorrectorSamplingParameters(**cases.sampling_ns(64, 1, M=M, greedy=False, one=False, cell=[10.86] * 3),
                                                        rng_mode="device", seed=5, use_hip_graph=use_graph and wrap)
        if repaint:
            from diffusion_for_multi_scale_molecular_dynamics_amd.generators.constrained_langevin_generator import \
                ConstrainedLangevinGenerator
            from diffusion_for_multi_scale_molecular_dynamics_amd.generators.sampling_constraint import SamplingConstraint
            spar.repaint_resampling_steps = 1
            g = torch.Generator().manual_seed(8)
            constraint = SamplingConstraint(elements=["Si"], constrained_relative_coordinates=torch.rand(20, 3, generator=g),
                                            constrained_atom_types=torch.zeros(20, dtype=torch.long))
            gen = ConstrainedLangevinGenerator(npar, spar, net, constraint)
        else:
            gen = LangevinGenerator(npar, spar, net)
        gen._prepare(cuda)
        if wrap:
            gen.axl_network = _RangeReportAt(net, float(gen.noise.time[k])).to(cuda)
        return gen, net

    gen, net = build(wrap=True)
    with torch.no_grad(), pytest.warns(UserWarning, match="f16 range"):
        got = gen.sample(B, cuda)
    assert gen.f16_range_fallbacks == flagged and net.edge_chain_precision == precision
    # the same run with the precision switched by hand, iteration by iteration (eager; replay == eager is tested above)
    ref, ref_net = build(wrap=False)
    with torch.no_grad():
        ref._begin_call(cuda)
        comp = ref.initialize(B, cuda)
        forces = torch.zeros_like(comp.X)
        for i in range(T - 1, -1, -1):
            in_f32 = i == k or (M > 0 and i == k + 1)       # predictor of k; correctors of k + 1 (time index k + 1: time[k])
            ref_net.edge_chain_precision = "f32" if in_f32 else precision
            comp = ref._iteration(comp, i, forces)
        if repaint:
            comp = ref._apply_constraint(comp, cuda)          # (what ConstrainedLangevinGenerator.sample does at the end)
        ref.check_status()
    assert torch.equal(got.A, comp.A) and torch.equal(got.X, comp.X)
    assert torch.isfinite(got.X).all() and (got.A == 0).all()
    # and an all-split run of the same seed differs from it in the last bits only (the two f32 iterations)
    plain, plain_net = build(wrap=False)
    with torch.no_grad():
        other = plain.sample(B, cuda)
    assert plain.f16_range_fallbacks == 0
    diff = (other.X - got.X + 0.5) % 1.0 - 0.5
    assert float(diff.norm() / got.X.norm()) < 1e-5


@pytest.mark.parametrize("use_graph", [True, False])
def test_f16_range_fallback_many_events(cuda, use_graph):
    """The watched loop under several reports in one call: the FIRST iteration (nothing queued before it), two consecutive ones
    (the second is flagged again when it is repeated behind the first's f32 pass), one in the middle and the LAST iteration
    (time index 0: nothing queued behind it) -- five iterations recomputed, and the result equals, bit for bit, the run with
    exactly those five computed by the f32 kernels."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    import cases
    import warnings
    T, B = 24, 4
    flagged = [T - 1, 15, 14, 7, 0]

    def build(wrap):
        torch.manual_seed(21)
        net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=2).to(cuda)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = NoiseParameters(**cases.noise_ns(T, **cases.LIN))
            spar = PredictorCorrectorSamplingParameters(**cases.sampling_ns(64, 1, M=0, greedy=False, one=False, cell=[10.86] * 3),
                                                        rng_mode="device", seed=5, use_hip_graph=use_graph and wrap)
        gen = LangevinGenerator(npar, spar, net)
        gen._prepare(cuda)
        if wrap:
            gen.axl_network = _RangeReportAt(net, gen.noise.time[flagged]).to(cuda)
        return gen, net

    gen, net = build(wrap=True)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = gen.sample(B, cuda)
    assert gen.f16_range_fallbacks == len(flagged)
    ref, ref_net = build(wrap=False)
    with torch.no_grad():
        ref._begin_call(cuda)
        comp = ref.initialize(B, cuda)
        forces = torch.zeros_like(comp.X)
        for i in range(T - 1, -1, -1):
            ref_net.edge_chain_precision = "f32" if i in flagged else "f16x3"
            comp = ref._iteration(comp, i, forces)
        ref.check_status()
    assert torch.equal(got.A, comp.A) and torch.equal(got.X, comp.X)


@pytest.mark.parametrize("precision", CHAIN_MODES)
@pytest.mark.parametrize("H,n_layers,M,with_residual", [(32, 1, 77, True), (64, 2, 500, False), (128, 3, 129, True),
                                                        (256, 5, 1500, True)])
def test_mlp_chain_rows_against_fp64(cuda, precision, H, n_layers, M, with_residual):
    """mdx_mlp_chain_rows (the edge-chain pipeline over the rows of a matrix; last layer linear; optional residual) against
    the fp64 evaluation of the same layer stack; M not a multiple of the 128-row tile."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    torch.manual_seed(H + n_layers)
    layers = [torch.nn.Linear(H, H) for _ in range(n_layers)]
    for layer in layers:
        with torch.no_grad():
            layer.weight.mul_(1.7)
    x = torch.randn(M, H)
    res = torch.randn(M, H) if with_residual else None
    y = x.double()
    for k, layer in enumerate(layers):
        y = y @ layer.weight.double().t() + layer.bias.double()
        if k < n_layers - 1:
            y = torch.nn.functional.silu(y)
    want = y + (res.double() if with_residual else 0)
    dev_layers = [layer.to(cuda) for layer in layers]
    pack = kernels.RowChainPack(dev_layers, precision)
    status = torch.zeros(1, dtype=torch.int32, device=cuda)
    got = kernels.mlp_chain_rows(pack, x.to(cuda), None if res is None else res.to(cuda), status=status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and torch.isfinite(got).all()
    tol = TOLERANCE[precision]
    assert _rel_l2(got, want) < tol, (precision, H, _rel_l2(got, want))
    row_err = ((got.double().cpu() - want.detach()).norm(dim=1) / want.detach().norm(dim=1).clamp(min=1e-30)).max()
    assert float(row_err) < 20 * tol, float(row_err)


@pytest.mark.parametrize("precision", CHAIN_MODES)
@pytest.mark.parametrize("H,n_nodes,deg", [(32, 40, 7), (128, 500, 25), (256, 900, 25), (256, 300, 90), (64, 70, 1)])
def test_edge_chain_piece_sums_against_fp64(cuda, precision, H, n_nodes, deg):
    """Message aggregation inside the edge chain (MDX_EGNN_MESSAGES_PIECE_SUMS) + mdx_segment_combine against the fp64
    segment sum / mean of the fp64 messages: ragged degrees (zero-degree nodes, segments longer than a 16-edge group, a
    32-edge wavefront and a 128-edge tile), edge count not a multiple of anything; and equal to mdx_segment_rows of the
    kernel's own messages up to summation order."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    g = torch.Generator().manual_seed(7 * H + n_nodes)
    n_in, D, n_msg, n_crd = 12, 6, 2, 1
    torch.manual_seed(H)
    lin0 = torch.nn.Linear(2 * n_in + 1, H)
    msg = [torch.nn.Linear(H, H) for _ in range(n_msg)]
    crd = [torch.nn.Linear(H, H) for _ in range(n_crd)]
    out = torch.nn.Linear(H, 1, bias=False)
    degree = torch.randint(0, 2 * deg + 1, (n_nodes,), generator=g)
    degree[0] = 0
    degree[n_nodes // 2] = 0
    degree[-1] = max(int(degree[-1]), 3)
    src = torch.repeat_interleave(torch.arange(n_nodes), degree)
    E = int(src.numel())
    dst = torch.randint(0, n_nodes, (E,), generator=g)
    edges = torch.stack([src, dst], 1)
    h = torch.randn(n_nodes, n_in, generator=g)
    coord = torch.rand(n_nodes, D, generator=g) * 2 - 1
    want_m, want_s = _chain_reference(lin0, msg, crd, out, n_in, h, coord, edges)
    want_sum = torch.zeros(n_nodes, H, dtype=torch.float64).index_add_(0, src, want_m.detach())
    want_mean = want_sum / degree.clamp(min=1).double()[:, None]

    mods = [m.to(cuda) for m in [lin0] + msg + crd + [out]]
    pack = kernels.EdgeChainPack(mods[0], mods[1:1 + n_msg], mods[1 + n_msg:-1], mods[-1], input_size=n_in, precision=precision)
    assert pack.piece_sums_ok
    w = mods[0].weight.detach()
    proj = torch.nn.functional.linear(h.to(cuda), torch.cat([w[:, :n_in], w[:, n_in:2 * n_in]], 0)).contiguous()
    coord_d, edges_d = coord.to(cuda).contiguous(), edges.to(cuda)
    offsets = (torch.cumsum(degree, 0) - degree).to(cuda)
    status = torch.zeros(1, dtype=torch.int32, device=cuda)
    pieces, scalar = kernels.egnn_edge_chain(pack, proj, coord_d, edges_d, status=status, piece_sums=True)
    got_mean = kernels.segment_combine(pieces, E, offsets, degree.to(cuda), True)
    got_sum = kernels.segment_combine(pieces, E, offsets, degree.to(cuda), False)
    messages, scalar2 = kernels.egnn_edge_chain(pack, proj, coord_d, edges_d, status=status, piece_sums=False)
    rows_mean = kernels.segment_rows(messages, offsets, degree.to(cuda), True)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    tol = TOLERANCE[precision]
    assert _rel_l2(got_mean, want_mean) < tol and _rel_l2(got_sum, want_sum) < tol, (_rel_l2(got_mean, want_mean), _rel_l2(got_sum, want_sum))
    assert _rel_l2(got_mean, rows_mean) < 1e-6
    assert torch.equal(scalar, scalar2)                             # the coordinate branch does not depend on the mode
    assert (got_sum[degree == 0] == 0).all()
    # [left | sums] in one pass: torch.cat([left, sums], 1), bit for bit
    left = torch.randn(n_nodes, H, generator=g).to(cuda)
    both = kernels.segment_combine(pieces, E, offsets, degree.to(cuda), True, left=left)
    assert both.shape == (n_nodes, 2 * H) and torch.equal(both, torch.cat([left, got_mean], dim=1))
    # the same and the coordinate update in ONE pass over the nodes (mdx_egnn_node_gather): messages bit for bit, coordinates
    # against fp64 and against the stand-alone kernel (another, equally fixed, summation order)
    for mean_c in (True, False):
        both2, coord_new = kernels.egnn_node_gather(pieces, E, offsets, degree.to(cuda), True, left, scalar, coord_d, edges_d, mean_c)
        assert torch.equal(both2, both)
        want_c = torch.zeros(n_nodes, D, dtype=torch.float64).index_add_(
            0, src, (coord.double()[src] - coord.double()[dst]) * scalar.double().cpu()[:, None])
        if mean_c:
            want_c = want_c / degree.clamp(min=1).double()[:, None]
        want_c = coord.double() + want_c
        assert _rel_l2(coord_new, want_c) < 1e-6
        alone = kernels.egnn_coord_aggregate(scalar, coord_d, edges_d, offsets, degree.to(cuda), mean_c)
        assert _rel_l2(coord_new, alone) < 1e-6 and torch.equal(coord_new[degree == 0], coord_d[degree == 0])
    sums_only, _ = kernels.egnn_node_gather(pieces, E, offsets, degree.to(cuda), False, None, scalar, coord_d, edges_d, True)
    assert torch.equal(sums_only, got_sum)
    node_err = ((got_mean.double().cpu() - want_mean).norm(dim=1) / want_mean.norm(dim=1).clamp(min=1e-30))[degree > 0].max()
    assert float(node_err) < 20 * tol, float(node_err)


@pytest.mark.parametrize("n_tiles", [1, 7, 8, 9, 15, 17, 255, 256, 257, 300, 513])
def test_edge_chain_tile_order_covers_every_tile(cuda, n_tiles):
    """The XCD-contiguous tile order of the edge chain (one eighth of the tiles per XCD, walked by that XCD's workgroups):
    every 128-edge tile is computed exactly once whatever the number of tiles is relative to 8, to the number of CUs and to
    their multiples -- rows mode, every edge's message and scalar against fp64; the last tile ragged; with and without a
    device-side edge count smaller than the list's capacity."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    H, n_in, D = 32, 6, 3
    E = 128 * (n_tiles - 1) + 37
    n_nodes = max(E // 9, 2)
    g = torch.Generator().manual_seed(n_tiles)
    torch.manual_seed(n_tiles)
    lin0, msg, crd = torch.nn.Linear(2 * n_in + 1, H), [torch.nn.Linear(H, H)], [torch.nn.Linear(H, H)]
    out = torch.nn.Linear(H, 1, bias=False)
    src = torch.sort(torch.randint(0, n_nodes, (E,), generator=g)).values
    dst = torch.randint(0, n_nodes, (E,), generator=g)
    edges = torch.stack([src, dst], 1)
    h = torch.randn(n_nodes, n_in, generator=g)
    coord = torch.rand(n_nodes, D, generator=g) * 2 - 1
    want_m, want_s = _chain_reference(lin0, msg, crd, out, n_in, h, coord, edges)
    mods = [m.to(cuda) for m in [lin0] + msg + crd + [out]]
    pack = kernels.EdgeChainPack(mods[0], mods[1:2], mods[2:3], mods[3], input_size=n_in, precision="f32")
    w = mods[0].weight.detach()
    proj = torch.nn.functional.linear(h.to(cuda), torch.cat([w[:, :n_in], w[:, n_in:2 * n_in]], 0)).contiguous()
    coord_d = coord.to(cuda).contiguous()
    for capacity in (E, E + 1000):
        edges_d = torch.zeros(capacity, 2, dtype=torch.int64, device=cuda)
        edges_d[:E] = edges.to(cuda)
        count = torch.tensor([E], dtype=torch.int64, device=cuda) if capacity > E else None
        got_m, got_s = kernels.egnn_edge_chain(pack, proj, coord_d, edges_d, n_edges_dev=count)
        row_err = (got_m[:E].double().cpu() - want_m).norm(dim=1) / want_m.norm(dim=1).clamp(min=1e-30)
        assert float(row_err.max()) < 1e-4, (capacity, int(row_err.argmax()) // 128, float(row_err.max()))
        assert _rel_l2(got_s[:E], want_s) < 1e-5


@pytest.mark.parametrize("H", [96, 30, 256])
def test_egnn_node_inputs_and_scores_against_torch(cuda, H):
    """mdx_egnn_node_inputs / mdx_egnn_scores against the expressions they replace in EGNNScoreNetwork: the embedded node
    features bit for bit (a one-hot input is a column pick: the same binary32 operations), the torus uplift and the score
    contraction to rounding (a different summation order over three / 2 n_k terms).  H = 30: the element-wise kernel; the
    others: one wavefront per node with 16-byte stores."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, positive_bloch_wave_vectors)
    g = torch.Generator().manual_seed(5)
    B, N, C = 7, 13, 2
    x = torch.rand(B, N, 3, generator=g).to(cuda)
    sigma = (torch.rand(B, 1, generator=g) * 0.5 + 0.01).to(cuda)
    a = torch.randint(0, C + 1, (B, N), generator=g).to(cuda)             # MASK class included
    emb = torch.nn.Linear(C + 2, H).to(cuda)
    kv = positive_bloch_wave_vectors(2, 3).to(cuda)
    z, h = kernels.egnn_node_inputs(x, kv, sigma.reshape(-1).contiguous(), a, emb.weight.detach(), emb.bias.detach())
    feats = torch.cat([sigma.repeat_interleave(N, dim=0), torch.nn.functional.one_hot(a.reshape(-1), C + 1).float()], dim=1)
    want_h = emb.bias.unsqueeze(0) + feats[:, :1] * emb.weight[:, 0].unsqueeze(0)
    for k in range(1, C + 2):
        want_h = torch.addcmul(want_h, feats[:, k:k + 1], emb.weight[:, k].unsqueeze(0))
    assert torch.equal(h, want_h.detach())
    # a second linear map of the same input in the same launch: P (W x + b) as (P W) x + P b
    P = torch.randn(2 * H, H, generator=g).to(cuda) / H ** 0.5
    w2 = (P.double() @ emb.weight.detach().double()).float().contiguous()
    b2 = (P.double() @ emb.bias.detach().double()).float().contiguous()
    z_b, h_b, proj = kernels.egnn_node_inputs(x, kv, sigma.reshape(-1).contiguous(), a, emb.weight.detach(), emb.bias.detach(),
                                              second=(w2, b2))
    assert torch.equal(z_b, z) and torch.equal(h_b, h)
    assert _rel_l2(proj, want_h.detach().double() @ P.double().t()) < 1e-6
    flat = x.reshape(B * N, 3)
    kr = ((2.0 * math.pi * flat)[:, None, :] * kv[None, :, :]).sum(dim=-1)
    want_z = torch.stack([kr.cos(), kr.sin()], dim=2).reshape(B * N, -1)
    assert (z - want_z).abs().max() < 2e-6
    x_hat = torch.randn(B * N, 2 * kv.shape[0], generator=g).to(cuda)
    gamma = EGNNScoreNetwork._projection_matrices(kv.cpu()).to(cuda)
    want_s = torch.einsum("ni,aij,nj->na", want_z.double(), gamma.double(), x_hat.double())
    got_s = kernels.egnn_scores(want_z.contiguous(), x_hat, kv)
    assert _rel_l2(got_s, want_s) < 1e-6


@pytest.mark.parametrize("precision", CHAIN_MODES)
@pytest.mark.parametrize("H,n_inner,M,with_residual", [(32, 0, 77, True), (64, 1, 500, False), (128, 2, 129, True),
                                                       (256, 4, 1000, True), (256, 0, 5, False)])
def test_node_mlp_rows_against_fp64(cuda, precision, H, n_inner, M, with_residual):
    """mdx_node_mlp_rows: the whole node MLP -- Linear(2H, H), SiLU, n_inner x (Linear(H, H), SiLU), Linear(H, H) -- and the
    residual on rows [h | agg], against fp64; ragged row counts (the first layer's two halves go through the matrix cores
    one after the other, the accumulators of the first waiting in registers)."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    torch.manual_seed(3 * H + n_inner)
    g = torch.Generator().manual_seed(M)
    layers = [torch.nn.Linear(2 * H, H)] + [torch.nn.Linear(H, H) for _ in range(n_inner + 1)]
    for layer in layers:
        with torch.no_grad():
            layer.weight.mul_(1.6)
    x = torch.randn(M, 2 * H, generator=g)
    y = x.double()
    for k, layer in enumerate(layers):
        y = y @ layer.weight.double().t() + layer.bias.double()
        if k + 1 < len(layers):
            y = y * torch.sigmoid(y)
    want = (x[:, :H].double() + y) if with_residual else y
    mods = [m.to(cuda) for m in layers]
    assert kernels.NodeMlpPack.supported(mods)
    pack = kernels.NodeMlpPack(mods, precision)
    status = torch.zeros(1, dtype=torch.int32, device=cuda)
    got = kernels.node_mlp_rows(pack, x.to(cuda).contiguous(), with_residual, status=status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    tol = TOLERANCE[precision]
    assert _rel_l2(got, want.detach()) < tol, _rel_l2(got, want.detach())
    row_err = ((got.double().cpu() - want.detach()).norm(dim=1) / want.detach().norm(dim=1).clamp(min=1e-30)).max()
    assert float(row_err) < 20 * tol, float(row_err)
    # with the next graph layer's projections appended: the same output bit for bit, and out @ P^T beside it
    proj_w = torch.randn(2 * H, H, generator=g) * (1.5 / H ** 0.5)
    pack_p = kernels.NodeMlpPack(mods, precision, next_projection=proj_w.to(cuda))
    assert pack_p.projects
    got2, proj = kernels.node_mlp_rows(pack_p, x.to(cuda).contiguous(), with_residual, status=status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and torch.equal(got2, got)
    want_p = want.detach() @ proj_w.double().t()
    assert _rel_l2(proj, want_p) < tol, _rel_l2(proj, want_p)
    # the two halves of the row as separate matrices (mdx_node_mlp_rows_split: what the EGNN layer passes): the same bits
    xd = x.to(cuda)
    got3, proj3 = kernels.node_mlp_rows(pack_p, xd[:, :H].contiguous(), with_residual, status=status, agg=xd[:, H:].contiguous())
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and torch.equal(got3, got) and torch.equal(proj3, proj)


# ---------------------------------------------------------------------------------------------------------------------
# round-5 regressions around the f16-range fallback
# ---------------------------------------------------------------------------------------------------------------------
def _unequal_width_net(cuda, precision="f16x3"):
    """message / coordinate width 64, node width 32: h and the messages differ in width, so the node MLP takes the
    Linear(2H -> H) library GEMM + kernels.RowChainPack path (models/egnn.py::_forward_edge_chain) and not NodeMlpPack."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    torch.manual_seed(23)
    net = EGNNScoreNetwork(EGNNScoreNetworkParameters(
        num_atom_types=1, n_layers=2, coordinate_hidden_dimensions_size=64, coordinate_n_hidden_dimensions=2,
        message_hidden_dimensions_size=64, message_n_hidden_dimensions=2, node_hidden_dimensions_size=32,
        node_n_hidden_dimensions=2, edges="radial_cutoff", radial_cutoff=7.5)).eval().to(cuda)
    net.edge_chain_precision = precision
    return net


@pytest.mark.parametrize("use_graph", [True, False])
def test_f16_range_fallback_with_the_row_chain_pack_keeps_the_captured_images(cuda, use_graph):
    """A captured iteration holds the split-f16 RowChainPack's image pointers in its kernel arguments; the f32 iteration of a
    fallback used to REPLACE (and free) that pack, so every replay after it read freed memory.  One image per precision is kept
    now: the watched hipGraph run with one flagged iteration equals, bit for bit, the run with exactly that iteration computed
    by the f32 kernels by hand."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    import cases
    import warnings
    T, B, k = 9, 5, 5

    def build(wrap):
        net = _unequal_width_net(cuda)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = NoiseParameters(**cases.noise_ns(T, **cases.LIN))
            spar = PredictorCorrectorSamplingParameters(**cases.sampling_ns(64, 1, M=0, greedy=False, one=False, cell=[10.86] * 3),
                                                        rng_mode="device", seed=5, use_hip_graph=use_graph and wrap)
        gen = LangevinGenerator(npar, spar, net)
        gen._prepare(cuda)
        if wrap:
            gen.axl_network = _RangeReportAt(net, float(gen.noise.time[k])).to(cuda)
        return gen, net

    gen, net = build(wrap=True)
    with torch.no_grad(), pytest.warns(UserWarning, match="f16 range"):
        got = gen.sample(B, cuda)
    assert gen.f16_range_fallbacks == 1
    layer = net.egnn.graph_layers[0]
    assert layer._node_chain[1] is not None and layer._node_mlp[1] is None, "the RowChainPack path did not run"
    assert set(layer._node_chain_kept) == {"f16x3", "f32"}, "one image per precision is kept"
    # graph replays after the fallback still run the split image (a second call replays the same captured iteration)
    with torch.no_grad(), pytest.warns(UserWarning, match="f16 range"):
        again = gen.sample(B, cuda)
    ref, ref_net = build(wrap=False)
    with torch.no_grad():
        outs = []
        for _ in range(2):
            ref._begin_call(cuda)
            comp = ref.initialize(B, cuda)
            forces = torch.zeros_like(comp.X)
            for i in range(T - 1, -1, -1):
                ref_net.edge_chain_precision = "f32" if i == k else "f16x3"
                comp = ref._iteration(comp, i, forces)
            ref.check_status()
            outs.append(comp)
    assert torch.equal(got.A, outs[0].A) and torch.equal(got.X, outs[0].X)
    assert torch.equal(again.A, outs[1].A) and torch.equal(again.X, outs[1].X)


class _OverflowAtTimeZeroIndex(_RangeReportAt):
    """As _RangeReportAt, and the forward's LOGITS come out non-finite as well -- what a real overflow in the node path gives."""

    def forward(self, batch, conditional=None):
        from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL, TIME
        out = super().forward(batch, conditional)
        if self.net.edge_chain_precision in ("f16x3", "f16x3_32x32"):
            hit = (batch[TIME][:1, 0:1] == self.time_values).any()
            out = AXL(A=torch.where(hit, torch.full_like(out.A, float("nan")), out.A), X=out.X, L=out.L)
        return out


@pytest.mark.parametrize("use_graph", [False, True])
def test_overflow_in_the_last_iteration_is_redone_without_a_stale_mask_report(cuda, use_graph):
    """An overflow in the LAST predictor step (time index 1 -> 0, every atom still MASKed, non-finite logits AND scores from the
    split-f16 attempt): the exact-f32 redo is clean and nothing the dropped attempt raised in either status word survives it --
    both loops; check_status() inside sample_from_noisy_composition() does not raise, every atom ends unmasked and finite.
    (With NaN logits this kernel, like torch.max in the reference, takes class 0: the MASK-at-last-step bit is not raised by
    such an attempt today; the eager loop clears it before the redo all the same, as the watched loop always did.)"""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    import cases
    import warnings
    T, B = 4, 5
    torch.manual_seed(21)
    net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=2).to(cuda)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = NoiseParameters(**cases.noise_ns(T, **cases.LIN))
        spar = PredictorCorrectorSamplingParameters(**cases.sampling_ns(64, 1, M=1, greedy=False, one=False, cell=[10.86] * 3),
                                                    rng_mode="device", seed=5, use_hip_graph=use_graph)
    gen = LangevinGenerator(npar, spar, net)
    gen._prepare(cuda)
    gen.axl_network = _OverflowAtTimeZeroIndex(net, float(gen.noise.time[0])).to(cuda)
    start = AXL(A=torch.ones(B, 64, dtype=torch.long, device=cuda), X=torch.rand(B, 64, 3, device=cuda),
                L=torch.tensor([10.86, 10.86, 10.86, 0, 0, 0], device=cuda).repeat(B, 1))      # every atom still MASKed
    with torch.no_grad(), pytest.warns(UserWarning, match="f16 range"):
        gen._begin_call(cuda)
        out = gen.sample_from_noisy_composition(start, 2, 0)          # check_status() inside: must not raise
    assert gen.f16_range_fallbacks >= 1 and (out.A == 0).all() and torch.isfinite(out.X).all()
    assert int(gen._status.item()) == 0 and int(net.graph_status.item()) == 0


@pytest.mark.parametrize("use_graph", [True, False])
def test_a_stale_range_bit_is_not_read_as_the_first_iterations_report(cuda, use_graph):
    """A range bit left in the network's status word by something outside the loop (the warm-up iterations before a capture, a
    caller stepping by hand) used to cost a spurious f32 iteration, a warning and lowered exponents."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    import cases
    import warnings
    torch.manual_seed(21)
    net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=2).to(cuda)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = NoiseParameters(**cases.noise_ns(5, **cases.LIN))
        spar = PredictorCorrectorSamplingParameters(**cases.sampling_ns(64, 1, M=1, greedy=False, one=False, cell=[10.86] * 3),
                                                    rng_mode="device", seed=5, use_hip_graph=use_graph)
    gen = LangevinGenerator(npar, spar, net)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        first = gen.sample(4, cuda)
        gen._call_counter = 0
        net.graph_status.bitwise_or_(_hip.STATUS_EGNN_F16_RANGE)          # stale: nothing of the next call has run
        second = gen.sample(4, cuda)
    assert gen.f16_range_fallbacks == 0
    assert torch.equal(first.X, second.X) and torch.equal(first.A, second.A)
    scales = [s for layer in net.egnn.graph_layers for s in layer._activation_scales.values()]
    assert scales and all(int(s.exponents.min()) == 6 for s in scales)


def test_activation_scale_state_can_be_reset(cuda):
    """The exponents are the one piece of state a fallback leaves behind (they only go down): begin_f16_range_fallback() zeroes
    the collected maxima, reset_f16_range() restores the defaults."""
    torch.manual_seed(21)
    net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=2).to(cuda)
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    X = torch.rand(3, 64, 3, device=cuda)
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.zeros(3, 64, dtype=torch.long, device=cuda), X=X,
                                        L=torch.tensor([10.86, 10.86, 10.86, 0, 0, 0], device=cuda).repeat(3, 1)),
             TIME: torch.full((3, 1), 0.5, device=cuda), NOISE: torch.full((3, 1), 0.1, device=cuda),
             CARTESIAN_FORCES: torch.zeros_like(X)}
    net.edge_chain_precision = "f32"
    with torch.no_grad():
        net(batch, conditional=False)
    scales = [s for layer in net.egnn.graph_layers for s in layer._activation_scales.values()]
    assert any(int(s.maxima.max()) != 0 for s in scales), "the exact-f32 kernels collect activation maxima"
    net.begin_f16_range_fallback()
    assert all(int(s.maxima.abs().max()) == 0 for s in scales)
    for s in scales:
        s.exponents.fill_(2)
    net.reset_f16_range()
    assert all(int(s.exponents.min()) == 6 and int(s.exponents.max()) == 6 for s in scales)


@pytest.mark.parametrize("precision", CHAIN_MODES)
@pytest.mark.parametrize("H,n_msg,n_crd,n_nodes,deg", [(32, 1, 1, 40, 7), (64, 2, 3, 300, 11), (128, 3, 2, 500, 25),
                                                        (256, 4, 5, 1200, 25), (256, 2, 2, 3, 2)])
def test_edge_chain_attention_gate_against_fp64(cuda, precision, H, n_msg, n_crd, n_nodes, deg):
    """The attention instantiations (ATT) of the chain kernel alone, against fp64: m_e <- m_e sigmoid(m_e . w_att + b_att)
    between the message and the coordinate layers (models/egnn.py:148-160), the gate's weight scaled so that its logits have a
    standard deviation of 1.5: the gate really varies between edges.  Ragged sorted edge list (degrees 0 .. 2 deg, not a multiple of the 128-edge tile), the
    in-kernel message sums combined per node against the fp64 segment sums of the GATED messages, the head's scalar against the
    fp64 coordinate MLP of the gated messages; a device-side edge count below the capacity leaves later rows untouched; the
    rows mode is refused for an attention chain (MDX_ERR_UNSUPPORTED: the gate is instantiated for the piece sums)."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    g = torch.Generator().manual_seed(2000 * H + n_nodes)
    n_in, D = 24, 6
    torch.manual_seed(H + n_msg + 100)
    lin0 = torch.nn.Linear(2 * n_in + 1, H)
    msg = [torch.nn.Linear(H, H) for _ in range(n_msg)]
    crd = [torch.nn.Linear(H, H) for _ in range(n_crd)]
    out = torch.nn.Linear(H, 1, bias=False)
    att = torch.nn.Linear(H, 1)
    with torch.no_grad():
        for layer in msg + crd:
            layer.weight.mul_(1.7)
    degree = torch.randint(0, 2 * deg, (n_nodes,), generator=g)
    degree[0] = 0
    degree[-1] = max(int(degree[-1]), 6)
    src = torch.repeat_interleave(torch.arange(n_nodes), degree)
    E = int(src.numel())
    dst = torch.randint(0, n_nodes, (E,), generator=g)
    edges = torch.stack([src, dst], 1)
    h = torch.randn(n_nodes, n_in, generator=g)
    coord = torch.rand(n_nodes, D, generator=g) * 2 - 1
    # fp64 reference: messages, gate, gated messages -> coordinate MLP -> head; segment sums of the gated messages
    f64, silu = torch.float64, torch.nn.functional.silu
    diff = coord.to(f64)[src] - coord.to(f64)[dst]
    x = torch.cat([h.to(f64)[src], h.to(f64)[dst], (diff ** 2).sum(1, keepdim=True)], dim=1)
    x = silu(x @ lin0.weight.to(f64).t() + lin0.bias.to(f64))
    for layer in msg:
        x = silu(x @ layer.weight.to(f64).t() + layer.bias.to(f64))
    with torch.no_grad():                       # spread the gate's logits to a standard deviation of 1.5: the gate must vary
        x = x.detach()
        att.weight.mul_(1.5 / float((x @ att.weight.to(f64).t()).std().clamp(min=1e-12)))
        logit = x @ att.weight.to(f64).t() + att.bias.to(f64)
    assert 1.0 < float(logit.std()) < 2.0
    x = x * torch.sigmoid(logit)
    y = x
    for layer in crd:
        y = silu(y @ layer.weight.to(f64).t() + layer.bias.to(f64))
    want_s = (y @ out.weight.to(f64).t()).reshape(-1)
    # (the head's scalar is a 256-term sum that cancels to a tenth or less of its terms in these random networks -- torch's own
    # binary32 evaluation of the largest case is 8e-6 from fp64 in relative terms -- so its error is measured against the sum of
    # the terms' magnitudes, the scale its rounding errors live on; the plain relative error gets a 20 x looser bar)
    want_s_scale = (y.abs() @ out.weight.to(f64).abs().t()).reshape(-1)
    want_sums = torch.zeros(n_nodes, H, dtype=f64).index_add_(0, src, x)

    mods = [m.to(cuda) for m in [lin0] + msg + crd + [out, att]]
    lin0_d, msg_d, crd_d, out_d, att_d = mods[0], mods[1:1 + n_msg], mods[1 + n_msg:-2], mods[-2], mods[-1]
    pack = kernels.EdgeChainPack(lin0_d, msg_d, crd_d, out_d, input_size=n_in, precision=precision, attention_layer=att_d)
    assert pack.att_w is not None and pack.piece_sums_ok
    w = lin0_d.weight.detach()
    proj = torch.nn.functional.linear(h.to(cuda), torch.cat([w[:, :n_in], w[:, n_in:2 * n_in]], 0)).contiguous()
    status = torch.zeros(1, dtype=torch.int32, device=cuda)
    coord_d, edges_d = coord.to(cuda).contiguous(), edges.to(cuda)
    pieces, got_s = kernels.egnn_edge_chain(pack, proj, coord_d, edges_d, status=status, piece_sums=True)
    degree_d = degree.to(cuda)
    offsets_d = (torch.cumsum(degree_d, 0) - degree_d).contiguous()
    got_sums = kernels.segment_combine(pieces, E, offsets_d, degree_d, False)
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and torch.isfinite(got_s).all() and torch.isfinite(got_sums).all()
    tol = TOLERANCE[precision]
    err_s, err_m = _rel_l2(got_s, want_s), _rel_l2(got_sums, want_sums)
    err_s_scaled = float((got_s.double().cpu() - want_s).norm() / want_s_scale.norm())
    assert err_s_scaled < tol and err_s < 20 * tol and err_m < tol, (precision, H, err_s_scaled, err_s, err_m)
    # ungated, the same chain gives something else (the gate is not a no-op in this test)
    plain = kernels.EdgeChainPack(lin0_d, msg_d, crd_d, out_d, input_size=n_in, precision=precision)
    _, plain_s = kernels.egnn_edge_chain(plain, proj, coord_d, edges_d, piece_sums=True)
    assert _rel_l2(plain_s, want_s) > 1e-2
    # a device-resident edge count below the capacity: the head's scalars beyond it are not written
    if E > 5:
        n_dev = torch.tensor([E - 5], dtype=torch.int64, device=cuda)
        _, s2 = kernels.egnn_edge_chain(pack, proj, coord_d, edges_d, n_edges_dev=n_dev, piece_sums=True)
        torch.cuda.synchronize()
        assert torch.equal(s2[:E - 5], got_s[:E - 5])
    # rows mode + attention: refused, not silently ungated
    with pytest.raises(_hip.MdxError):
        kernels.egnn_edge_chain(pack, proj, coord_d, edges_d, piece_sums=False)


EGNN_FUZZ = max(1, int(os.environ.get("MDX_FUZZ", "1")))


@pytest.mark.parametrize("seed", range(24 * EGNN_FUZZ))
def test_random_egnn_configurations_gpu_against_the_cpu_module(cuda, seed):
    """Seeded random EGNNScoreNetwork configurations -- 1 - 3 spatial dimensions, 1 - 3 graph layers, message / coordinate / node
    widths 8 ... 256 (equal or not: the chain's zero-padding), 1 - 4 hidden layers each, every combination of attention / tanh /
    normalize / residual, sum or mean aggregations, fully connected or radial-cutoff graphs, 1 - 3 atom types, 2 - 40 atoms -- on
    the HIP path in the exact-f32 and the split-f16 arithmetic against THIS package's module evaluated in binary64 on the CPU
    (the module the reference-made fixtures pin: net_egnn_variants, net_egnn_options_wide, low_dimensions; its radial graphs from
    the oracle's edge list).  Scores within max(2e-5, 3 x floor) rel-L2 of the binary64 answer, floor = the same module in
    binary32 on the CPU against its binary64 self (what binary32 costs this random-init network); logits close."""
    import warnings
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE, NOISY_AXL_COMPOSITION,
                                                                              TIME)
    rng = np.random.default_rng(9000 + seed)
    d = int(rng.choice([1, 2, 3], p=[0.2, 0.2, 0.6]))
    nat = int(rng.integers(1, 4))
    equal = bool(rng.random() < 0.5)
    widths = [int(rng.choice([8, 16, 32, 48, 64, 96, 128, 256]))] * 3 if equal else [int(rng.choice([8, 16, 32, 48, 64, 128])) for _ in range(3)]
    radial = bool(rng.random() < 0.6)
    rc = float(rng.uniform(2.5, 4.0))
    p = EGNNScoreNetworkParameters(
        spatial_dimension=d, num_atom_types=nat, n_layers=int(rng.integers(1, 4)),
        message_hidden_dimensions_size=widths[0], message_n_hidden_dimensions=int(rng.integers(1, 5)),
        coordinate_hidden_dimensions_size=widths[1], coordinate_n_hidden_dimensions=int(rng.integers(1, 5)),
        node_hidden_dimensions_size=widths[2], node_n_hidden_dimensions=int(rng.integers(1, 5)),
        attention=bool(rng.random() < 0.5), tanh=bool(rng.random() < 0.5), normalize=bool(rng.random() < 0.5),
        residual=bool(rng.random() < 0.7), coords_agg=str(rng.choice(["mean", "sum"])), message_agg=str(rng.choice(["mean", "sum"])),
        edges="radial_cutoff" if radial else "fully_connected", radial_cutoff=rc if radial else None)
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = EGNNScoreNetwork(p).eval()
    B, N = int(rng.integers(1, 6)), int(rng.integers(2, 41))
    g = torch.Generator().manual_seed(seed)
    lengths = torch.rand(B, d, generator=g) * 4.0 + 2.2 * rc + 0.5           # cells the radial graph accepts (> 2.2 x cutoff)
    L = torch.cat([lengths, torch.zeros(B, d * (d + 1) // 2 - d)], dim=1)
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, nat + 1, (B, N), generator=g), X=torch.rand(B, N, d, generator=g), L=L),
             TIME: torch.rand(B, 1, generator=g), NOISE: torch.rand(B, 1, generator=g) * 0.2 + 0.01,
             CARTESIAN_FORCES: torch.zeros(B, N, d)}
    with torch.no_grad():
        # the CPU module takes the oracle's edge list (this package's radius graph is the HIP kernel: no host tensors)
        builder = nets.oracle_edge_builder if radial else None
        net64 = EGNNScoreNetwork(p, edge_builder=builder).eval()
        net64.load_state_dict(net.state_dict())
        net32 = EGNNScoreNetwork(p, edge_builder=builder).eval()
        net32.load_state_dict(net.state_dict())
        net64 = net64.double()
        batch64 = {k: (AXL(A=v.A, X=v.X.double(), L=v.L.double()) if k == NOISY_AXL_COMPOSITION else v.double()) for k, v in batch.items()}
        want = net64(batch64, conditional=False)
        scale = float(want.X.norm())
        # what binary32 arithmetic costs THIS random network: the same module in float32 on the CPU against its binary64 self
        floor = float((net32(batch, conditional=False).X.double() - want.X).norm()) / max(scale, 1e-30)
        net = net.to(cuda)
        on_device = {k: (AXL(*[t.to(cuda) for t in v]) if k == NOISY_AXL_COMPOSITION else v.to(cuda)) for k, v in batch.items()}
        for precision in ("f32", "f16x3"):
            net.edge_chain_precision = precision
            got = net(on_device, conditional=False)
            net.check_status()
            err = float((got.X.cpu().double() - want.X).norm()) / max(scale, 1e-30)
            assert err < max(2e-5, 3.0 * floor), (seed, precision, err, floor, p)
            finite = torch.isfinite(want.A)
            assert torch.allclose(got.A.cpu().double()[finite], want.A[finite], rtol=2e-4, atol=2e-5), (seed, precision)
            assert torch.equal(torch.isfinite(got.A.cpu()), finite)
            # where the fused edge chain applies (widths up to 256, at most eight layers, an attention gate that fits) it RAN
            fused = [layer._edge_chain_pack() is not None for layer in net.egnn.graph_layers]
            assert all(layer._chain[1] is not None and layer._chain[1].precision == precision
                       for layer, ok in zip(net.egnn.graph_layers, fused) if ok)
        print(f"FUSED {sum(fused)} of {len(fused)} layers; widths {widths}; d {d}; radial {radial}")


@pytest.mark.parametrize("name,N,B,edges,rc", [("one_atom_fully_connected", 1, 3, "fully_connected", None), ("one_atom_radial", 1, 3, "radial_cutoff", 2.0),
                                               ("no_pair_within_the_cutoff", 4, 2, "radial_cutoff", 0.05), ("two_atoms_one_structure", 2, 1, "radial_cutoff", 2.5)])
def test_egnn_on_graphs_without_edges(cuda, name, N, B, edges, rc):
    """Degenerate graphs -- a single atom, no pair within the cutoff, one structure of two atoms: the forward is finite (an atom
    without neighbours is not moved by the EGNN: its score is exactly zero), and the sampler around it runs, the hipGraph replay
    equal to the eager launches bit for bit, every atom unmasked at the end."""
    import warnings
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE, NOISY_AXL_COMPOSITION,
                                                                              TIME)
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    p = EGNNScoreNetworkParameters(num_atom_types=1, n_layers=2, coordinate_hidden_dimensions_size=32, coordinate_n_hidden_dimensions=2,
                                   message_hidden_dimensions_size=32, message_n_hidden_dimensions=2, node_hidden_dimensions_size=32,
                                   node_n_hidden_dimensions=2, edges=edges, radial_cutoff=rc)
    torch.manual_seed(1)
    net = EGNNScoreNetwork(p).eval().to(cuda)
    g = torch.Generator().manual_seed(2)
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.zeros(B, N, dtype=torch.long, device=cuda), X=torch.rand(B, N, 3, generator=g).to(cuda),
                                        L=torch.tensor([6.0, 6.0, 6.0, 0, 0, 0.0]).repeat(B, 1).to(cuda)),
             TIME: torch.rand(B, 1, generator=g).to(cuda), NOISE: (torch.rand(B, 1, generator=g) * 0.2).to(cuda),
             CARTESIAN_FORCES: torch.zeros(B, N, 3, device=cuda)}
    with torch.no_grad():
        out = net(batch, conditional=False)
    net.check_status()
    assert out.X.shape == (B, N, 3) and torch.isfinite(out.X).all() and torch.isfinite(out.A[..., :-1]).all()
    if name != "two_atoms_one_structure":
        assert float(out.X.abs().max()) == 0.0
    else:
        assert float(out.X.norm()) > 0.0
    results = {}
    for use_graph in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            spar = PredictorCorrectorSamplingParameters(number_of_atoms=N, num_atom_types=1, number_of_samples=B, number_of_corrector_steps=1,
                                                        use_fixed_lattice_parameters=True, cell_dimensions=[6.0] * 3, rng_mode="device",
                                                        seed=3, use_hip_graph=use_graph)
            gen = LangevinGenerator(NoiseParameters(total_time_steps=3, sigma_min=1e-4, sigma_max=0.2, schedule_type="linear"), spar, net)
            with torch.no_grad():
                results[use_graph] = gen.sample(B, cuda)
    assert torch.equal(results[False].X, results[True].X) and torch.equal(results[False].A, results[True].A)
    assert (results[True].A == 0).all() and ((results[True].X >= 0) & (results[True].X < 1)).all()
