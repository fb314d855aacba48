"""GPU parity of the BENCHMARKED network shape against the REFERENCE's own outputs.

bench.py's C3 / C4 / C5 lines run the reference's production EGNN -- 4 graph layers, 256 wide, 4 hidden layers per MLP,
radius graph at rc = 7.5 (experiments/.../Si_2x2x2/config_diffusion_egnn.yaml:44-60) -- through
egnn_edge_chain_kernel<256, PREC, 2> and <256, PREC, 3>.  The fixtures net_egnn_c3 / traj_egnn_c3_{top,bottom} hold what the
reference computed at that shape (tests/golden/make_golden.py::golden_c3_shape; weights from tests/formula_weights.py), so
these tests hold the fused HIP forward, in BOTH arithmetic modes of the MFMA kernels, and the sampler steps built on it
against the reference itself -- not against the product's own module:

  * network forward: scores within 1e-5 rel-L2 (north_star's tolerance), logits close, MASK logit -inf;
  * every predictor / corrector step of the T = 1000 schedule's first two and last two indices, started from the
    composition the reference recorded and fed the reference's draws: atom types exact, coordinates within 1e-5 (torus);
  * the same four indices in free run.
"""
import numpy as np
import pytest
import torch

import cases
import nets
from conftest import load_golden, torus_rel_l2
from oracle import reference_sampler as RS
from test_generator_gpu import _pkg, _replayed
import teacher_forced

pytestmark = pytest.mark.gpu

MODES = ["f32", "f16x3", None]          # exact-f32 MFMA | split-f16 MFMA (the default) | per-layer library GEMMs


def _batch(g, device):
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    t = lambda k: torch.from_numpy(g[k]).to(device)      # noqa: E731
    return {NOISY_AXL_COMPOSITION: AXL(A=t("A"), X=t("X"), L=t("L")), TIME: t("time"), NOISE: t("noise"),
            CARTESIAN_FORCES: torch.zeros(g["X"].shape, device=device)}


@pytest.mark.parametrize("precision", MODES)
def test_c3_network_forward_against_reference(cuda, precision):
    g = load_golden("net_egnn_c3.npz")
    net = nets.egnn_c3_net(1).to(cuda)
    net.edge_chain_precision = precision
    with torch.no_grad():
        out = net(_batch(g, cuda), conditional=False)
    net.check_status()
    assert torch.isinf(out.A[..., -1]).all() and (out.A[..., -1] < 0).all()
    ref = g["out_X"].astype(np.float64)
    err = np.linalg.norm(out.X.cpu().numpy() - ref) / np.linalg.norm(ref)
    assert err < 1e-5, f"{precision}: scores rel-L2 {err:.2e} against the reference"
    np.testing.assert_allclose(out.A.cpu().numpy()[..., :-1], g["out_A"][..., :-1], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out.L.cpu().numpy(), g["out_L"], rtol=1e-5, atol=1e-7)
    if precision is not None:        # the fused chain really ran: every graph layer holds a packed image for this mode
        assert all(layer._chain[1] is not None and layer._chain[1].precision == precision
                   for layer in net.egnn.graph_layers)


@pytest.mark.parametrize("precision", MODES)
@pytest.mark.parametrize("fixture,num_atom_types", [("net_egnn_c3_wide", 1), ("net_egnn_c4", 2), ("net_egnn_c3_live", 1),
                                                    ("net_egnn_c4_live", 2)])
def test_production_network_forward_on_sampler_like_inputs(cuda, fixture, num_atom_types, precision):
    """The production network on the inputs a sampler meets (tests/golden/make_golden.py::golden_c3_wide): 32 structures at
    sigma = 1e-4 ... 0.2 -- uniform-random ones, the diamond sites of Si 2x2x2 displaced by sigma z (the end of a trajectory),
    half-MASKed ones -- and 8 structures of the two-atom-type network of configs[3]: scores <= 1e-5 rel-L2 against the
    reference's output over the batch, per structure max(2e-5, the reference's own floor) and 1e-5 of the rms score in absolute
    terms (tests/teacher_forced.py::forward_check explains the floor: near the diamond sites the score nearly cancels);
    logits close."""
    g = load_golden(fixture + ".npz")
    net = nets.egnn_c3_net(num_atom_types, scale=nets.LIVE_SCALE if fixture.endswith("_live") else 1.0).to(cuda)
    net.edge_chain_precision = precision
    with torch.no_grad():
        out = net(_batch(g, cuda), conditional=False)
    net.check_status()
    assert torch.isinf(out.A[..., -1]).all() and (out.A[..., -1] < 0).all()
    err, worst, where, floor = teacher_forced.forward_check(out.X.cpu().numpy(), g, f"{fixture} / {precision}")
    print(f"{fixture} / {precision}: batch {err:.2e}, worst structure {worst:.2e} (#{where}, sigma "
          f"{float(g['noise'][where, 0]):.1e}); the reference's own distance from binary64 over the batch: {floor:.2e}")
    np.testing.assert_allclose(out.A.cpu().numpy()[..., :-1], g["out_A"][..., :-1], rtol=1e-4, atol=1e-5)
    if precision is not None:
        assert all(layer._chain[1] is not None and layer._chain[1].precision == precision
                   for layer in net.egnn.graph_layers)


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_c3_full_size_batch_properties(cuda, precision):
    """BASELINE configs[2] at its FULL batch (512 structures x 64 atoms, ~800 k edges: what bench.py runs) held to the reference
    through properties that do not depend on the size:
      * the reference's eight structures of net_egnn_c3, scattered among 504 random ones, come out as the REFERENCE computed
        them alone (<= 1e-5 rel-L2 on the scores; logits close) -- a structure's output does not depend on its batch, and the
        512-structure launch geometry (6 000+ workgroups, XCD tile order, the capacity-sized edge list, piece rows shared by
        neighbouring nodes) is the one that is benchmarked;
      * the same eight inside the batch and alone on the HIP path agree to 1e-5 (3e-6 measured: the message sums are grouped
        by 16-edge pieces whose boundaries move with the structure's place in the edge list);
      * permuting the atoms of every structure permutes scores and logits (<= 1e-5: the edge order changes);
      * two forwards of the same batch are bit-identical (no atomics, fixed summation order)."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    g = load_golden("net_egnn_c3.npz")
    net = nets.egnn_c3_net(1).to(cuda)
    net.edge_chain_precision = precision
    B, N = 512, 64
    gen = torch.Generator().manual_seed(512)
    X = torch.rand(B, N, 3, generator=gen)
    A = torch.randint(0, 2, (B, N), generator=gen)
    L = torch.from_numpy(g["L"][:1]).repeat(B, 1)
    noise = torch.rand(B, 1, generator=gen) * 0.4 + 0.01
    time = torch.rand(B, 1, generator=gen)
    where = torch.tensor([3, 64, 65, 200, 255, 256, 400, 511])               # both ends of workgroup tiles and of the batch
    for k, b in enumerate(where.tolist()):
        X[b], A[b], L[b] = torch.from_numpy(g["X"][k]), torch.from_numpy(g["A"][k]), torch.from_numpy(g["L"][k])
        noise[b], time[b] = torch.from_numpy(g["noise"][k]), torch.from_numpy(g["time"][k])

    def forward(X, A, L, time, noise):
        batch = {NOISY_AXL_COMPOSITION: AXL(A=A.to(cuda), X=X.to(cuda), L=L.to(cuda)), TIME: time.to(cuda),
                 NOISE: noise.to(cuda), CARTESIAN_FORCES: torch.zeros(X.shape, device=cuda)}
        with torch.no_grad():
            out = net(batch, conditional=False)
        net.check_status()
        return out

    full = forward(X, A, L, time, noise)
    ref = g["out_X"].astype(np.float64)
    got = full.X[where.to(cuda)].cpu().numpy()
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert err < 1e-5, f"{precision}: the reference's structures inside a 512-structure batch: scores rel-L2 {err:.2e}"
    np.testing.assert_allclose(full.A[where.to(cuda)].cpu().numpy()[..., :-1], g["out_A"][..., :-1], rtol=1e-4, atol=1e-5)
    alone = forward(X[where], A[where], L[where], time[where], noise[where])
    assert float((alone.X - full.X[where.to(cuda)]).norm() / alone.X.norm()) < 1e-5
    again = forward(X, A, L, time, noise)
    assert torch.equal(again.X, full.X) and torch.equal(again.A, full.A)
    perm = torch.stack([torch.randperm(N, generator=gen) for _ in range(B)])
    rows = torch.arange(B)[:, None]
    permuted = forward(X[rows, perm], A[rows, perm], L, time, noise)
    want = full.X[rows.to(cuda), perm.to(cuda)]
    assert float((permuted.X - want).norm() / want.norm()) < 1e-5
    wl, gl = full.A[rows.to(cuda), perm.to(cuda)][..., :-1], permuted.A[..., :-1]
    assert float((gl - wl).norm() / wl.norm()) < 1e-5


def test_c3_full_size_sampler_graph_replay_equals_eager(cuda):
    """The benchmarked iteration at its full size (512 structures, the production-size EGNN, device RNG, M = 2) on a four-index
    schedule: captured into a hipGraph and replayed, against the same iterations launched eagerly and against the two-call
    radius graph (edge list sized to the real edge count after a host read) -- the same bits; every atom ends unmasked and
    inside the unit cell."""
    import warnings
    P = _pkg()
    noise_kw, sampling_kw, _ = cases.C3_SHAPE
    noise_kw = dict(noise_kw, total_time_steps=4)
    outs = {}
    for mode in ("eager", "graph", "two_call"):
        net = nets.egnn_c3_net(1).to(cuda)
        if mode == "two_call":
            net.static_edge_list_max_fraction = 0.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = P["Noise"](**noise_kw)
            spar = P["Sampling"](**dict(sampling_kw), rng_mode="device", seed=77, use_hip_graph=mode == "graph")
        gen = P["Langevin"](npar, spar, net)
        with torch.no_grad():
            out = gen.sample(512, cuda)
        assert gen.f16_range_fallbacks == 0
        outs[mode] = (out.A.cpu().numpy(), out.X.cpu().numpy())
    for mode in ("graph", "two_call"):
        assert np.array_equal(outs["eager"][0], outs[mode][0]), mode
        assert np.array_equal(outs["eager"][1].view(np.int32), outs[mode][1].view(np.int32)), mode
    assert (outs["eager"][0] == 0).all() and np.isfinite(outs["eager"][1]).all()       # every atom unmasked, on the torus
    assert (outs["eager"][1] >= 0).all() and (outs["eager"][1] < 1).all()


def _generator(cuda, precision, shape=None, **extra):
    import warnings
    P = _pkg()
    noise_kw, sampling_kw, netf = shape or cases.C3_SHAPE
    skw = dict(sampling_kw)
    skw.update(extra)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar, spar = P["Noise"](**noise_kw), P["Sampling"](**skw)
    net = netf(None).to(cuda)
    net.edge_chain_precision = precision
    return P["Langevin"](npar, spar, net), spar


C3_C4_TRAJECTORIES = ["traj_egnn_c3_top", "traj_egnn_c3_bottom", "traj_egnn_c4_top", "traj_egnn_c4_mid",
                      "traj_egnn_c3_live", "traj_egnn_c4_live_bottom"]


def _shape_of(name):
    """configs[2] (Si, one atom type) or configs[3] (SiGe: two atom types, greedy sampling + one transition per step), with
    the scale-1 or the "live" formula weights"""
    return cases.shape_of(name)


@pytest.mark.parametrize("precision", ["f32", "f16x3", "f16x3_32x32"])
@pytest.mark.parametrize("name", C3_C4_TRAJECTORIES)
def test_c3_teacher_forced_steps(cuda, name, precision, record_property):
    """Every recorded step of the reference at production width, from the reference's composition with the reference's draws:
    atom types exact, coordinates <= 1e-5 (torus) -- and, because those two cannot see the network at these step sizes (module
    docstring of tests/teacher_forced.py), the NETWORK'S OUTPUT of every step against the reference's recorded
    `model_predictions_i`: scores <= 1e-5 rel-L2 per step, logits close.  36 forwards x 4 structures at 4 x 256 x 4, sigma from
    0.2 down to sigma_min, the two-atom-type network of configs[3] included.  The two `live` fixtures run the network with
    formula weights at 2 x the default range, where the hidden layers matter to the output (a 0.1 % error in one 256 x 256
    matrix moves the scores by 1.4e-4: test_live_fixtures_see_the_hidden_layers); traj_egnn_c4_live_bottom ends at time index 0
    with C = 3: the one-transition rule is off in the last predictor step and no MASK may remain
    (src/.../generators/langevin_generator.py:601-604,616-620 -- the status word read by check_status())."""
    g = load_golden(name + ".npz")
    gen, spar = _generator(cuda, precision, shape=_shape_of(name))
    gen.noise_source = _replayed(g)
    records = teacher_forced.run(gen, spar, g, cuda)
    record_property("steps", teacher_forced.summary(records))
    print(f"{name} / {precision}: {teacher_forced.summary(records)}")
    teacher_forced.check(records, f"{name} / {precision}")
    if name == "traj_egnn_c4_live_bottom":
        mask = spar.num_atom_types
        assert (g["pred_composition_i_A"][0] == mask).any() and not (g["pred_composition_im1_A"][-1] == mask).any()
        # the first predictor (index 2) obeys the one-transition rule, the last one (index 1) does not
        changed = [(g["pred_composition_i_A"][k] != g["pred_composition_im1_A"][k]).sum(axis=1) for k in range(2)]
        assert changed[0].max() <= 1 and changed[1].max() > 1


@pytest.mark.parametrize("name,factor,logit_factor,message", [
    ("traj_egnn_c3_top", 0.0, 1.0, "scores"), ("traj_egnn_c3_bottom", 0.0, 1.0, "scores"),
    ("traj_egnn_c4_mid", 0.0, 1.0, "scores"), ("traj_egnn_c4_mid", 1.0, 0.0, "atom types|logits"),
    ("traj_egnn_c3_top", 1.0 + 3e-5, 1.0, "scores")])
def test_teacher_forced_steps_fail_with_a_wrong_network(cuda, name, factor, logit_factor, message):
    """NEGATIVE CONTROL.  The same teacher-forced replay with the network's scores zeroed (or its logits zeroed, or its scores
    off by 3e-5 relative) must FAIL -- and it is the network-output assertion that catches it at the 1e-5 level: with zeroed
    scores the atom types still match, every predictor output is still within 1e-5, and a corrector output is off by < 1e-3
    (asserted here: it is the blindness the output assertion exists for)."""
    g = load_golden(name + ".npz")
    gen, spar = _generator(cuda, "f16x3", shape=_shape_of(name))
    gen.axl_network = nets.ScaledScore(gen.axl_network, factor, logit_factor)
    gen.noise_source = _replayed(g)
    records = teacher_forced.run(gen, spar, g, cuda)
    with pytest.raises(AssertionError, match=message):
        teacher_forced.check(records, name)
    if logit_factor == 1.0:
        # what the step outputs alone see of a zeroed score: atom types nothing; the PREDICTOR's coordinates nothing (g^2 s / sigma
        # = 1.4e-6 of |X|); the correctors' coordinates 4e-4 .. 1e-3 at the top and middle of the schedule (their step
        # eps_i s / sigma grows with sigma_i^2) and nothing at the bottom -- i.e. X <= 1e-5 holds the score to a few per cent in
        # the correctors of large sigma and not at all elsewhere
        assert all(r["a_equal"] for r in records)
        assert all(r["x_err"] < 1e-5 for r in records if r["kind"] == "pred" or "bottom" in name), \
            [(r["kind"], r["k"], r["x_err"]) for r in records]
        assert all(r["x_err"] < 5e-3 for r in records)
        assert min(r["score_err"] for r in records) > (0.99 if factor == 0.0 else 2e-5)


@pytest.mark.parametrize("which", ["message", "coordinate", "node"])
def test_live_fixtures_see_the_hidden_layers(cuda, which):
    """NEGATIVE CONTROL for the hidden layers: ONE 256 x 256 matrix of the first graph layer off by 0.1 % -- the teacher-forced
    replay of traj_egnn_c3_live must FAIL on the network output (with the scale-1 weights of traj_egnn_c3_top the same error
    moves the scores by 3e-6 and passes: asserted too, it is why the live fixtures exist)."""
    def worst(name):
        g = load_golden(name + ".npz")
        gen, spar = _generator(cuda, "f16x3", shape=_shape_of(name))
        layer = gen.axl_network.egnn.graph_layers[0]
        mlp = dict(message=layer.message_mlp, coordinate=layer.coord_mlp, node=layer.node_mlp)[which]
        with torch.no_grad():
            mlp[2].weight.mul_(1.001)
        gen.noise_source = _replayed(g)
        records = teacher_forced.run(gen, spar, g, cuda)
        return records, max(max(r["score_err"] for r in records), 0.0 if all(r["logits_close"] for r in records) else 1.0)

    records, live = worst("traj_egnn_c3_live")
    with pytest.raises(AssertionError, match="network output"):
        teacher_forced.check(records, "perturbed")
    assert live > 2e-5
    if which != "node":          # (the node MLP feeds the logits and the next layers' messages, not this layer's coordinates)
        _, plain = worst("traj_egnn_c3_top")
        assert plain < 1.5e-5 and plain < live / 3, (plain, live)       # scale 1: at most marginal (8e-6 + 3e-6 .. 9e-6)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("name", C3_C4_TRAJECTORIES)
def test_c3_free_run_against_reference(cuda, name, precision):
    """The same indices without teacher forcing (each step starts from the HIP path's own previous output).  Holds the LOOP --
    step order, draw order, index handling, the status word -- to the reference; like the per-step coordinates it cannot see the
    network at these step sizes (that is test_c3_teacher_forced_steps' network-output assertion)."""
    g = load_golden(name + ".npz")
    gen, spar = _generator(cuda, precision, shape=_shape_of(name))
    gen.noise_source = _replayed(g)
    start = RS.AXL(A=torch.from_numpy(g["start_A"]).to(cuda), X=torch.from_numpy(g["start_X"]).to(cuda),
                   L=torch.from_numpy(g["start_L"]).to(cuda))
    with torch.no_grad():
        out = gen.sample_from_noisy_composition(start, int(g["start_index"]), int(g["end_index"]))
    gen.check_status()
    assert gen.noise_source.inner.exhausted()
    assert np.array_equal(out.A.cpu().numpy(), g["final_A"])
    err = torus_rel_l2(out.X.cpu().numpy(), g["final_X"])
    assert err < 1e-5, f"{name} / {precision}: final rel-L2 {err:.2e}"


def test_c3_arithmetic_modes_agree_over_many_iterations(cuda):
    """Error accumulation: 24 sampler iterations (72 network forwards) from the top of the T = 1000 schedule with the
    production-size EGNN (4 x 256 x 4, formula weights), device RNG (the same draws in every mode), B = 6: the split-f16 kernels
    (both MFMA shapes) and the library-GEMM path against the exact-f32 MFMA kernels -- atom types equal, coordinates within
    1e-5 rel-L2 on the torus (observed 1e-7), no range fallback.

    The graph here is FULLY CONNECTED on purpose.  With `edges: radial_cutoff` the network is a discontinuous function of the
    coordinates -- an edge appears or disappears when a pair distance crosses the cutoff -- so two fp32 evaluations that agree to
    4e-8 after three iterations can sit 1.7e-5 apart after 24 because ONE pair crossed 7.5 A in one run and not in the other
    (measured on this very set-up: {f32 MFMA, split-f16 16x16} and {split-f16 32x32, library GEMMs} each agree within 1.4e-7 and
    differ from each other by 1.68e-5).  That is a property of the model (the reference on other hardware has it too), not of
    an arithmetic mode; parity with the radius graph is therefore held per step (teacher-forced) and over short free runs
    above, and the accumulation of rounding differences is measured where the function is smooth."""
    import warnings
    P = _pkg()
    noise_kw, sampling_kw, _ = cases.C3_SHAPE
    outs = {}
    for precision in ("f32", "f16x3", "f16x3_32x32", None):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = P["Noise"](**noise_kw)
            spar = P["Sampling"](**dict(sampling_kw), rng_mode="device", seed=77)      # (eager: the fully connected edge list is built on the host)
        from formula_weights import fill_with_formula
        net = fill_with_formula(nets.egnn_net(1, "fully_connected", None, hidden=256, n_layers=4, n_hidden=4)).to(cuda)
        net.edge_chain_precision = precision
        gen = P["Langevin"](npar, spar, net)
        with torch.no_grad():
            gen._prepare(cuda)
            gen._begin_call(cuda)
            start = gen.initialize(6, cuda)
            out = gen.sample_from_noisy_composition(start, 1000, 976)
        outs[precision] = (out.A.cpu().numpy(), out.X.cpu().numpy())
        assert np.isfinite(outs[precision][1]).all() and gen.f16_range_fallbacks == 0
        if precision is not None:
            assert all(layer._chain[1] is not None and layer._chain[1].precision == precision for layer in net.egnn.graph_layers)
    for precision in ("f16x3", "f16x3_32x32", None):
        assert np.array_equal(outs[precision][0], outs["f32"][0])
        err = torus_rel_l2(outs[precision][1], outs["f32"][1])
        assert err < 1e-5, f"{precision} vs f32 after 24 iterations: {err:.2e}"


@pytest.mark.parametrize("precision", ["f32", "f16x3", "f16x3_32x32"])
@pytest.mark.parametrize("name", ["attention", "normalize_tanh", "sum_noresidual", "all_duplicates_kept"])
def test_egnn_option_variants_on_the_gpu_against_reference(cuda, name, precision):
    """Every option of the reference's E_GCL (models/egnn.py:128-135, 148-160, 234-264) on the hand-written path: HIP radius
    graph (one sorted list for either drop_duplicate_edges setting), the fused MFMA edge chain with the attention gate inside
    it, normalize / tanh in the per-node kernel that adds the coordinate updates up -- against the reference's forward: scores
    within 1e-5 rel-L2 (or the reference's own distance from the exact answer where that is larger), all three arithmetic
    modes, and the chain really ran in every graph layer."""
    from test_host_cpu import variant_case, variant_net, variant_tolerance
    g = load_golden("net_egnn_variants.npz")
    net, batch = variant_case(g, name, variant_net(name), device=cuda)
    net.edge_chain_precision = precision
    with torch.no_grad():
        out = net(batch, conditional=False)
    net.check_status()
    ref = g[f"{name}/out_X"].astype(np.float64)
    err = np.linalg.norm(out.X.cpu().numpy() - ref) / np.linalg.norm(ref)
    tol = variant_tolerance(g, name)          # 1e-5, or the reference's own distance from the exact answer where that is larger
    assert err < tol, f"{name} / {precision}: {err:.2e} (tolerance {tol:.2e})"
    np.testing.assert_allclose(out.A.cpu().numpy()[..., :-1], g[f"{name}/out_A"][..., :-1], rtol=1e-4, atol=1e-5)
    for layer in net.egnn.graph_layers:
        assert layer._chain[1] is not None and layer._chain[1].precision == precision, "the fused edge chain did not run"
        assert (layer._chain[1].att_w is not None) == layer.attention


@pytest.mark.parametrize("precision", ["f32", "f16x3", "f16x3_32x32"])
@pytest.mark.parametrize("name", ["template_1d", "attention_256", "normalize_128", "default_widths", "unequal_48_96"])
def test_egnn_options_at_kernel_widths_on_the_gpu_against_reference(cuda, name, precision):
    """The options at the widths 128 and 256 (tests/golden/make_golden.py::golden_egnn_options_wide): the reference's shipped
    1-D template (normalize=True, hidden 128, spatial dimension 1: coordinates of dimension D = 2 in the chain), attention + tanh
    at 256 (the ATT instantiation of the production-width kernel), attention + normalize with sum aggregations at 128; and
    NARROW / UNEQUAL widths, which the chain runs zero-padded to its next width (kernels.EdgeChainPack): the reference's default
    hyper-parameters (message 16, node 32, coordinate 32 -> chain width 32) and message 48 / coordinate 96 / node 64 with
    attention + tanh (-> 128) -- scores <= 1e-5 against the reference's output, logits close, the fused chain in every layer."""
    from test_host_cpu import wide_option_case
    g = load_golden("net_egnn_options_wide.npz")
    net, batch = wide_option_case(g, name, device=cuda)
    net.edge_chain_precision = precision
    with torch.no_grad():
        out = net(batch, conditional=False)
    net.check_status()
    ref = g[f"{name}/out_X"].astype(np.float64)
    err = np.linalg.norm(out.X.cpu().numpy() - ref) / np.linalg.norm(ref)
    floor = np.linalg.norm(ref - g[f"{name}/out_X_fp64"]) / np.linalg.norm(ref)
    print(f"{name} / {precision}: {err:.2e} (the reference's own distance from binary64: {floor:.2e})")
    assert err < 1e-5, f"{name} / {precision}: {err:.2e}"
    np.testing.assert_allclose(out.A.cpu().numpy()[..., :-1], g[f"{name}/out_A"][..., :-1], rtol=1e-4, atol=1e-5)
    for layer in net.egnn.graph_layers:
        assert layer._chain[1] is not None and layer._chain[1].precision == precision, "the fused edge chain did not run"


@pytest.mark.parametrize("network_from", ["model_block", "checkpoint_hyper_parameters"])
def test_cli_with_the_experiment_configuration_and_a_checkpoint(cuda, tmp_path, network_from):
    """sample_diffusion end to end on the reference's experiment configuration (experiments/.../Si_2x2x2/config_diffusion_egnn.yaml:
    the `model: score_network:` block and the `diffusion_sampling` noise / sampling blocks, T shortened), the network loaded
    from a Lightning-style checkpoint whose keys are the reference module's (`axl_network.` prefix): same files as the reference
    writes, and the samples equal a direct LangevinGenerator run with the same seed (sub-batches of 3 + 2).
    checkpoint_hyper_parameters: the REFERENCE's flow (src/.../sample_diffusion.py:191-205) -- the configuration holds `noise:`
    and `sampling:` only (experiments/.../Si_1x1x1/config_sample_T=1000.yaml) and the network is rebuilt from the hyper-parameters
    pickled inside the checkpoint, read without Lightning and without the reference package (utils/lightning_checkpoint.py)."""
    import warnings
    import yaml
    from diffusion_for_multi_scale_molecular_dynamics_amd import sample_diffusion
    P = _pkg()
    score_network = dict(architecture="egnn", num_atom_types=1, n_layers=4, coordinate_hidden_dimensions_size=256,
                         coordinate_n_hidden_dimensions=4, coords_agg="mean", message_hidden_dimensions_size=256,
                         message_n_hidden_dimensions=4, message_agg="mean", node_hidden_dimensions_size=256,
                         node_n_hidden_dimensions=4, attention=False, normalize=False, residual=True, tanh=False,
                         edges="radial_cutoff", radial_cutoff=7.5)
    noise = dict(total_time_steps=3, sigma_min=0.0001, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8)
    sampling = dict(algorithm="predictor_corrector", num_atom_types=1, number_of_atoms=64, sample_batchsize=3,
                    spatial_dimension=3, number_of_corrector_steps=2, one_atom_type_transition_per_step=False,
                    atom_type_greedy_sampling=False, atom_type_transition_in_corrector=False, number_of_samples=5,
                    record_samples=False, use_fixed_lattice_parameters=True, cell_dimensions=[10.86, 10.86, 10.86],
                    rng_mode="device", seed=123, use_hip_graph=True)
    net = nets.egnn_c3_net(1)
    if network_from == "model_block":
        (tmp_path / "config.yaml").write_text(yaml.safe_dump(dict(noise=noise, sampling=sampling, elements=["Si"],
                                                                  model=dict(score_network=score_network))))
        torch.save({"state_dict": {"axl_network." + k: v for k, v in net.state_dict().items()}}, tmp_path / "last_model.ckpt")
    else:
        (tmp_path / "config.yaml").write_text(yaml.safe_dump(dict(noise=noise, sampling=sampling)))
        parameters = net._hyper_params
        assert {k: getattr(parameters, k) for k in score_network} == score_network
        nets.write_lightning_style_checkpoint(tmp_path / "last_model.ckpt", net, parameters)
    sample_diffusion.main(["--config", str(tmp_path / "config.yaml"), "--checkpoint", str(tmp_path / "last_model.ckpt"),
                           "--output", str(tmp_path / "out"), "--device", "cuda"])
    samples = torch.load(tmp_path / "out" / "samples.pt", weights_only=False)
    axl = samples["original_axl"]
    assert samples["cartesian_positions"].shape == (5, 64, 3) and axl.A.shape == (5, 64) and axl.L.shape == (5, 6)
    assert (axl.A == 0).all() and ((axl.X >= 0) & (axl.X < 1)).all()
    assert torch.allclose(samples["cartesian_positions"].cpu(), axl.X.cpu() * 10.86)
    assert (tmp_path / "out" / "config_backup.yaml").exists() and (tmp_path / "out" / "console.log").exists()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gen = P["Langevin"](P["Noise"](**noise), P["Sampling"](**sampling), nets.egnn_c3_net(1).to(cuda))
    with torch.no_grad():
        direct = torch.cat([gen.sample(3, cuda).X, gen.sample(2, cuda).X])
    assert torch.equal(direct.cpu(), axl.X.cpu())


def test_cli_with_the_shipped_si_1x1x1_sampling_configuration(cuda, tmp_path):
    """The reference's own sampling run, as shipped: experiments/training_and_sampling_generative_models/inputs_and_scripts/Si_1x1x1/
    config_sample_T=1000.yaml (restated below key for key: noise + sampling + the metrics / elements / spatial_dimension / oracle
    blocks it carries, NO model block) with a Lightning-style checkpoint of the network that directory trains
    (config_diffusion_egnn.yaml:44-60: EGNN 4 x 256 x 4, fully connected) -- the command line of draw_samples.sh.  The 1000-step
    job runs in the reference's mode (host draws, eager launches: the file sets no fast-mode key), writes samples.pt and the
    recorded trajectories, and says that the oracle block is not evaluated."""
    import yaml
    from diffusion_for_multi_scale_molecular_dynamics_amd import sample_diffusion
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    shipped = dict(
        noise=dict(total_time_steps=1000, sigma_min=0.0001, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8),
        sampling=dict(algorithm="predictor_corrector", num_atom_types=1, sample_batchsize=16, spatial_dimension=3,
                      number_of_corrector_steps=2, number_of_atoms=8, number_of_samples=16, record_samples=True,
                      use_fixed_lattice_parameters=True, cell_dimensions=[5.43, 5.43, 5.43]),
        metrics=dict(compute_energies=True, compute_structure_factor=True, structure_factor_max_distance=5.0),
        elements=["Si"], spatial_dimension=3, oracle=dict(name="lammps", sw_coeff_filename="Si.sw"))
    (tmp_path / "config_sample.yaml").write_text(yaml.safe_dump(shipped))
    parameters = EGNNScoreNetworkParameters(num_atom_types=1, n_layers=4, coordinate_hidden_dimensions_size=256,
                                            coordinate_n_hidden_dimensions=4, coords_agg="mean", message_hidden_dimensions_size=256,
                                            message_n_hidden_dimensions=4, message_agg="mean", node_hidden_dimensions_size=256,
                                            node_n_hidden_dimensions=4, attention=False, normalize=False, residual=True, tanh=False,
                                            edges="fully_connected")
    torch.manual_seed(21)
    network = EGNNScoreNetwork(parameters)
    nets.write_lightning_style_checkpoint(tmp_path / "last_model.ckpt", network, parameters)
    sample_diffusion.main(["--config", str(tmp_path / "config_sample.yaml"), "--checkpoint", str(tmp_path / "last_model.ckpt"),
                           "--output", str(tmp_path / "samples"), "--device", "cuda"])
    samples = torch.load(tmp_path / "samples" / "samples.pt", weights_only=False)
    axl = samples["original_axl"]
    assert samples["cartesian_positions"].shape == (16, 8, 3) and axl.A.shape == (16, 8) and axl.L.shape == (16, 6)
    assert (axl.A == 0).all() and ((axl.X >= 0) & (axl.X < 1)).all() and torch.isfinite(axl.X).all()
    assert torch.allclose(samples["cartesian_positions"].cpu(), axl.X.cpu() * 5.43)
    trajectories = torch.load(tmp_path / "samples" / "trajectories.pt", weights_only=False)
    # (corrector steps are recorded only with record_samples_corrector_steps, which the file does not set: langevin_generator.py:74)
    assert len(trajectories["predictor_step"]) == 1000 and "corrector_step" not in trajectories
    assert trajectories["noise"].time.shape == (1000,) and trajectories["sampling_parameters"]["number_of_samples"] == 16
    log = (tmp_path / "samples" / "console.log").read_text()
    assert "energies.pt is not" in log and (tmp_path / "samples" / "config_backup.yaml").exists()
