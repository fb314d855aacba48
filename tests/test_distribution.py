"""Whole trajectories, held to the reference DISTRIBUTIONALLY.

Bit parity over a whole job is impossible between any two implementations (the MLP template's map is
chaotic, the radial cutoff makes the EGNN discontinuous), and the mode `value` is measured in -- device Philox, hardware exp2 /
sin / cos in the persistent MLP kernel, split-f16 MFMA products and a hipGraph loop for the EGNN -- is pinned to the oracle over
4 .. 24 iterations only.  The missing link is statistical: do complete jobs in that mode sample the distribution the REFERENCE
samples?  The reference's own measure for that question is the two-sample Kolmogorov-Smirnov distance
(src/.../metrics/kolmogorov_smirnov_metrics.py:7-75).

tests/golden/dist_*.npz (tests/golden/make_distributions.py, reference runs in the build container) hold, per scalar of
tests/distribution_stats.py, the quantile table of the reference's pooled final structures, the reference-vs-reference KS distances
(every seed against the pool of the others; pooled halves against each other) and the KS distance of deliberately WRONG samplers
(score zeroed or scaled, correctors dropped, ...) to the pool.  The cases:

  dist_mlp_c2            BASELINE configs[1]: the random-init MLP template, T = 1000, 1024 structures x 16 seeds
  dist_mlp_well          the same job, the template's weights making it a known periodic well: 3 % of the score's weight is seen
  dist_analytic          the reference's AnalyticalScoreNetwork as a plugin: the per-step kernels' case with 3 % power
  dist_egnn_rc           small radial-cutoff EGNN (score x 100), hipGraph loop, both MFMA modes
  dist_egnn_c3_wide      configs[2]'s production EGNN 4 x 256 x 4 (score x 150): the benchmarked kernel instantiation
  dist_egnn_repaint      the reference's ConstrainedLangevinGenerator: the free atoms around 32 pinned ones
  dist_egnn_types[_greedy]  two atom types, Gumbel-drawn and configs[3]'s greedy + one-transition settings

A sampler passes when

  * every one of its calls (the reference's batch per seed) is within MARGIN x the largest leave-one-out distance of the
    reference's own seeds, for every scalar;
  * its pooled sample is within MARGIN x the largest pooled-halves distance;
  * (networks that are not permutation equivariant) the same for the per-atom coordinate marginals: the largest and the mean
    over the 3 N marginals against the reference's largest and largest-mean;
  * no atom is left MASKED.

CPU part: the fixtures themselves (the wrong samplers FAIL the criterion: the check has teeth) and the CPU oracle in its Philox
mode on the MLP job.  GPU part (-m gpu): the product's fast paths.
"""
import numpy as np
import pytest
import torch

import cases
import distribution_stats as DS
import nets
from conftest import load_golden
from oracle import reference_sampler as RS

MARGIN = 1.5          # x the reference's own largest seed-to-seed distance (16 / 6 seeds: the largest of a few draws)
POOLED = ("pair", "nn", "x", "y", "z")


def thresholds(g, n_calls=None):
    """Per scalar: (limit for one call, limit for the pooled sample of n_calls calls); per-atom marginals: (limit of the largest,
    of the mean).  The pooled calibration -- halves of the reference's S seeds against each other -- is a comparison of S / 2
    calls with S / 2; a pooled sample of P calls against the table of S is noisier or quieter by sqrt((1/P + 1/S) / (4/S)),
    the factor the KS statistic of two samples scales with."""
    scalars = [str(k) for k in g["scalars"]]
    atoms = [k for k in scalars if k.startswith("atom")]
    S = len(g["seeds"])
    size = np.sqrt((S / (n_calls or S // 2) + 1.0) / 4.0)
    lim = {k: (MARGIN * g[f"leave_one_out/{k}"].max(), MARGIN * size * g[f"half_split/{k}"].max()) for k in scalars_of(g)}
    if atoms:
        loo = np.stack([g[f"leave_one_out/{k}"] for k in atoms])          # [marginals, seeds]
        lim["atoms"] = (MARGIN * loo.max(), MARGIN * loo.mean(0).max())
    return lim, atoms


TYPED = ("pair_same", "pair_diff", "nn_same")


def scalars_of(g):
    """the pooled scalars a fixture is judged on: the ones it names (`judged`), else the five common ones, `disp` when it names
    sites, `to_pinned` for a repaint run, the type-aware ones for several atom types"""
    if "judged" in g.files:
        return tuple(str(k) for k in g["judged"])
    return (POOLED + (("disp",) if "sites" in g.files else ()) + (("to_pinned",) if "pinned_sites" in g.files else ()) +
            (TYPED if "type_fraction/per_seed" in g.files else ()))


def type_fraction_limits(g):
    """(mean of the reference's seeds, allowed distance of one call from it): the fraction of atoms ending as type 0 is one
    number per call, held to MARGIN x the largest distance of a reference seed from the mean of the others."""
    f = g["type_fraction/per_seed"]
    loo = np.array([abs(f[i] - np.delete(f, i).mean()) for i in range(f.size)])
    return float(f.mean()), MARGIN * float(loo.max())


def measure(g, calls):
    """calls: list of X [B, N, 3] (one per sample() call), or of (X, A [B, N]) for a fixture with several atom types
    -> rows (what, measured KS distance, limit)."""
    lim, atoms = thresholds(g, len(calls))
    pooled = scalars_of(g)
    table = {k: g[f"table/{k}"] for k in list(pooled) + atoms}
    rows = []
    typed = "type_fraction/per_seed" in g.files
    per_call = [DS.statistics(c[0] if typed else c, per_atom=bool(atoms), sites=g["sites"] if "sites" in g.files else None,
                              pinned=g["pinned_sites"] if "pinned_sites" in g.files else None, types=c[1] if typed else None)
                for c in calls]
    if typed:
        mean, limit = type_fraction_limits(g)
        for c, (_, A) in enumerate(calls):
            rows.append((f"call {c}: fraction of type 0", abs(float((np.asarray(A) == 0).mean()) - mean), limit))
    for c, st in enumerate(per_call):
        for k in pooled:
            rows.append((f"call {c}: {k}", DS.ks_to_table(st[k], table[k]), lim[k][0]))
        if atoms:
            d = np.array([DS.ks_to_table(st[k], table[k]) for k in atoms])
            rows.append((f"call {c}: largest per-atom marginal", float(d.max()), lim["atoms"][0]))
            rows.append((f"call {c}: mean per-atom marginal", float(d.mean()), lim["atoms"][1]))
    if len(calls) > 1:
        for k in pooled:
            rows.append((f"pooled: {k}", DS.ks_to_table(np.concatenate([st[k] for st in per_call]), table[k]), lim[k][1]))
    return rows


def judge(g, calls):
    """The violated criteria (empty = passes)."""
    return [f"{what} {d:.4f} > {limit:.4f}" for what, d, limit in measure(g, calls) if d > limit]


def judge_confirmed(g, draw):
    """judge() with a controlled false-alarm rate.  The criterion compares ~40 KS distances per case with 1.5 x the largest of
    6 .. 16 reference leave-one-out distances: a CORRECT sampler exceeds one of them in a few per cent of the cases (seen in
    round 5: changing the last bit of the torus uplift re-drew every sample, and 2 of 22 cases came out at 1.004 x and 1.15 x a
    limit).  So a first set of calls that violates a limit is not yet a failure: a SECOND, independent set (the generator's
    next calls: fresh Philox streams) must then pass the whole criterion on its own -- every per-call limit and the pooled
    limits.  (The first set is not pooled into it: conditional on holding the outlier that raised the alarm its pooled
    statistic is biased upwards.)  A sampler that is wrong fails again -- the recorded wrong samplers of the fixtures sit at
    2 .. 20 x the limits -- while a false alarm of probability p becomes p^2.
    draw() -> a list of calls as judge() takes them.  Returns the violations (empty = passes)."""
    first = draw()
    bad = judge(g, first)
    if not bad:
        return []
    import warnings
    warnings.warn(f"distribution check: first set of calls violated {bad}; drawing a confirmation set")
    again = judge(g, draw())
    return (bad + again) if again else []


def probe_fails(g, probe):
    """Would the recorded wrong sampler `probe` have failed the criterion (its first call alone; its pooled calls where the
    fixture holds several)?"""
    calls = int(g["probe_calls"]) if "probe_calls" in g.files else 1
    lim, atoms = thresholds(g, calls)
    bad = any(float(g[f"probe/{probe}/{k}"]) > lim[k][0] for k in scalars_of(g))
    if "type_fraction/per_seed" in g.files:
        mean, limit = type_fraction_limits(g)
        bad = bad or abs(float(g[f"probe_type_fraction/{probe}"]) - mean) > limit
    if calls > 1:
        bad = bad or any(float(g[f"probe_pooled/{probe}/{k}"]) > lim[k][1] for k in scalars_of(g))
    if atoms:
        d = np.array([float(g[f"probe/{probe}/{k}"]) for k in atoms])
        bad = bad or d.max() > lim["atoms"][0] or d.mean() > lim["atoms"][1]
    return bad


@pytest.mark.parametrize("fixture,caught,missed", [
    ("dist_mlp_c2.npz", ["zero_score", "no_corrector"], ["score_x0.9", "sigma_max_0.2"]),
    ("dist_mlp_well.npz", ["zero_score", "score_x0.9", "score_x0.97", "no_corrector", "sigma_min_1e-3"], []),
    ("dist_egnn_rc.npz", ["zero_score", "score_x0.5", "no_corrector"], []),
    ("dist_egnn_c3_wide.npz", ["zero_score", "score_x0.5", "score_x0.9", "no_corrector"], []),
    ("dist_egnn_repaint.npz", ["zero_score", "score_x0.5", "no_corrector", "no_repaint"], []),
    ("dist_egnn_types.npz", ["zero_score", "uniform_types", "logits_x0.5", "other_type_update"], []),
    ("dist_egnn_types_greedy.npz", ["zero_score", "other_type_update"], ["uniform_types", "logits_x0.5"]),
    ("dist_analytic.npz", ["zero_score", "score_x0.9", "score_x0.97", "no_corrector", "sigma_min_1e-2"], ["sigma_max_0.2"])])
def test_the_criterion_has_teeth(fixture, caught, missed):
    """The reference's own wrong samplers against the criterion: a zeroed score, a halved score and a run without correctors
    are rejected; what the criterion cannot see at this sample size is listed too (a 10 % error of the MLP's score, a 20 %
    smaller sigma_max: inside the seed-to-seed spread of 1024 structures; under greedy type sampling the size of the logits:
    the argmax does not move) -- the check guards against gross errors of the fast
    mode, not against percent-level ones.  And the tables' own resolution is fine: a seed against the table is no further than
    against the exact pool of the other seeds."""
    g = load_golden(fixture)
    for probe in caught:
        assert probe_fails(g, probe), probe
    for probe in missed:
        assert not probe_fails(g, probe), probe
    for k in scalars_of(g):
        assert g[f"seed_vs_table/{k}"].max() <= g[f"leave_one_out/{k}"].max() * 1.05 + 1.0 / 2048


def analytic_case(g):
    """(noise kwargs, sampling kwargs, network) of tests/golden/dist_analytic.npz"""
    n_atoms, kmax, sigma_d, T, sigma_min, sigma_max = g["settings"]
    net = nets.GaussianWellScoreNetwork(g["sites"], float(sigma_d), int(kmax))
    return (cases.noise_ns(int(T), sigma_min=float(sigma_min), sigma_max=float(sigma_max)), cases.sampling_ns(int(n_atoms), 1), net)


def test_gaussian_well_network_against_reference_forward():
    """The tests' restatement of the reference's AnalyticalScoreNetwork against the reference's own forward (six noise levels
    from 1e-4 to 0.25)."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    g = load_golden("dist_analytic.npz")
    _, _, net = analytic_case(g)
    x, sigma = torch.from_numpy(g["forward/X"]), torch.from_numpy(g["forward/sigma"])
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.zeros(x.shape[:2], dtype=torch.long), X=x,
                                        L=torch.tensor([5.43, 5.43, 5.43, 0, 0, 0.0]).repeat(x.shape[0], 1)),
             TIME: torch.zeros(x.shape[0], 1), NOISE: sigma, CARTESIAN_FORCES: torch.zeros_like(x)}
    with torch.no_grad():
        out = net(batch, conditional=False)
    ref = g["forward/out_X"].astype(np.float64)
    assert np.linalg.norm(out.X.numpy() - ref) / np.linalg.norm(ref) < 1e-5
    assert np.array_equal(out.A.numpy(), g["forward/out_A"])


def test_oracle_samples_the_analytic_target(oracle):
    """The CPU oracle (Philox draws) on the analytic case, 4 calls of 1024 structures: the sampler contracts the uniform start
    onto the sites with the reference's final width -- a case where a 3 % error of the score's weight is REJECTED
    (test_the_criterion_has_teeth)."""
    g = load_golden("dist_analytic.npz")
    noise_kw, sampling_kw, net = analytic_case(g)
    npar, spar = cases.as_objects(noise_kw, sampling_kw)
    torch.set_num_threads(8)
    calls = [RS.OracleLangevinGenerator(npar, spar, net, noise=RS.PhiloxNoise(616, call)).sample(int(g["batch"])).X for call in range(4)]
    assert judge(g, calls) == []


def periodic_well_case(g):
    """(noise kwargs, sampling kwargs, network) of tests/golden/dist_mlp_well.npz: configs[1]'s job around the MLP template with
    the weights of cases.periodic_well_mlp_state"""
    amplitude, offset = g["well"]
    net = nets.mlp_net(8, 1)
    state = cases.periodic_well_mlp_state(g["sites"], amplitude=float(amplitude), offset=float(offset),
                                          reference_shapes={k: v.shape for k, v in net.state_dict().items()})
    net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    noise_kw, sampling_kw = mlp_c2_parameters()
    return noise_kw, sampling_kw, net


def test_periodic_well_mlp_against_reference_forward():
    """The product's MLP module with the constructed weights against the REFERENCE's module with the same weights (fixture) and
    against the closed form -amplitude sin(2 pi (x - site)): the two SiLUs are in their linear regime to 2e-6."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    g = load_golden("dist_mlp_well.npz")
    _, _, net = periodic_well_case(g)
    x = torch.from_numpy(g["forward/X"])
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.zeros(x.shape[:2], dtype=torch.long), X=x,
                                        L=torch.tensor([5.43, 5.43, 5.43, 0, 0, 0.0]).repeat(x.shape[0], 1)),
             TIME: torch.from_numpy(g["forward/time"]), NOISE: torch.from_numpy(g["forward/sigma"]),
             CARTESIAN_FORCES: torch.zeros_like(x)}
    with torch.no_grad():
        out = net(batch, conditional=False)
    assert np.abs(out.X.numpy() - g["forward/out_X"]).max() < 1e-6
    closed = -float(g["well"][0]) * np.sin(2 * np.pi * (g["forward/X"].astype(np.float64) - g["sites"][None]))
    assert np.abs(out.X.numpy() - closed).max() < 3e-6
    assert np.array_equal(out.A.numpy(), g["forward/out_A"])


def test_oracle_samples_the_periodic_well(oracle):
    """The CPU oracle (Philox draws) on the periodic-well job, 4 calls of 1024 structures."""
    g = load_golden("dist_mlp_well.npz")
    noise_kw, sampling_kw, net = periodic_well_case(g)
    npar, spar = cases.as_objects(noise_kw, sampling_kw)
    torch.set_num_threads(8)
    calls = [RS.OracleLangevinGenerator(npar, spar, net, noise=RS.PhiloxNoise(717, call)).sample(int(g["batch"])).X for call in range(4)]
    assert judge(g, calls) == []


def mlp_c2_parameters():
    noise_kw = cases.noise_ns(1000, sigma_min=1e-4, sigma_max=0.25)
    return noise_kw, cases.sampling_ns(8, 1)


def test_oracle_philox_mode_samples_the_reference_distribution(oracle):
    """The CPU oracle with the DEVICE random-number specification (Philox) on BASELINE configs[1]'s whole job (MLP template,
    T = 1000, 1024 structures): same distribution as the reference's torch-CPU draws."""
    g = load_golden("dist_mlp_c2.npz")
    noise_kw, sampling_kw = mlp_c2_parameters()
    npar, spar = cases.as_objects(noise_kw, sampling_kw)
    net = nets.load_fixture_weights(nets.mlp_net(8, 1), load_golden("net_mlp_c1.npz"))
    torch.set_num_threads(8)
    out = RS.OracleLangevinGenerator(npar, spar, net, noise=RS.PhiloxNoise(515, 0)).sample(int(g["batch"]))
    assert (out.A == 0).all()
    assert judge(g, [out.X]) == []


@pytest.mark.gpu
def test_fused_mlp_sampler_samples_the_reference_distribution(cuda):
    """BASELINE configs[1] in the mode bench.py's C2 `value` is measured in: the persistent fused kernel (one launch per
    trajectory, pre-drawn device Philox noise, hardware exp2 / sin / cos), 16 calls of 1024 structures."""
    from test_generator_gpu import _pkg
    import warnings
    P = _pkg()
    g = load_golden("dist_mlp_c2.npz")
    noise_kw, sampling_kw = mlp_c2_parameters()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**noise_kw)
        spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=9090, fused_score_network=True)
    net = nets.load_fixture_weights(nets.mlp_net(8, 1), load_golden("net_mlp_c1.npz")).to(cuda)
    gen = P["Langevin"](npar, spar, net)

    def draw():
        calls = []
        with torch.no_grad():
            for _ in range(len(g["seeds"])):
                out = gen.sample(int(g["batch"]), cuda)
                assert (out.A == 0).all()
                calls.append(out.X.cpu().numpy())
        return calls

    assert judge_confirmed(g, draw) == []
    # and the per-step path with the PyTorch forward in a hipGraph (the plugin path), fewer calls
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=9191, use_hip_graph=True)
    gen = P["Langevin"](npar, spar, net)
    gen_steps = gen

    def draw_steps():
        with torch.no_grad():
            return [gen_steps.sample(int(g["batch"]), cuda).X.cpu().numpy() for _ in range(2)]

    assert judge_confirmed(g, draw_steps) == []


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["product", "padded_family", "generic_folded", "generic_layer_by_layer", "per_step_graph"])
def test_mlp_samplers_sample_the_periodic_well(cuda, variant):
    """configs[1]'s job around the MLP template with known weights (a periodic well of amplitude 0.3 around the diamond sites:
    the reference's own runs with the score x 0.9 are rejected per call) through every sampler that takes an MLP: the
    persistent fused kernel in its register-resident exact family (what bench.py's C2 `value` runs), the padded family, the
    generic kernel with the folded and with the layer-by-layer forward -- one launch per trajectory, pre-drawn device Philox
    noise, hardware exp2 / sin / cos -- and the per-step kernels around the PyTorch module in a hipGraph.  16 calls of 1024
    structures (the per-step path: 4)."""
    from test_generator_gpu import _pkg
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
    import warnings
    P = _pkg()
    g = load_golden("dist_mlp_well.npz")
    noise_kw, sampling_kw, net = periodic_well_case(g)
    fused = variant != "per_step_graph"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**noise_kw)
        spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=3131, fused_score_network=fused, use_hip_graph=not fused)
    gen = P["Langevin"](npar, spar, net.to(cuda))
    gen.fused_sampler_options = {"padded_family": _hip.MLP_SAMPLE_PADDED_FAMILY, "generic_folded": _hip.MLP_SAMPLE_GENERIC_KERNEL,
                                 "generic_layer_by_layer": _hip.MLP_SAMPLE_GENERIC_KERNEL | _hip.MLP_SAMPLE_UNFOLDED}.get(variant, 0)

    def draw():
        calls = []
        with torch.no_grad():
            for _ in range(len(g["seeds"]) if fused else 4):
                out = gen.sample(int(g["batch"]), cuda)
                assert (out.A == 0).all()
                calls.append(out.X.cpu().numpy())
        return calls

    assert judge_confirmed(g, draw) == []


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_egnn_graph_loop_samples_the_reference_distribution(cuda, precision):
    """A small radial-cutoff EGNN (hidden 32, N = 64, T = 100 of configs[2]'s schedule, M = 2) whose coordinate score is
    multiplied by 100 (a plugin around the network, on both sides: with the bare random-init network the final distribution is
    uniform whatever the sampler does), in the mode bench.py's C3 `value` is measured in: device Philox, HIP radius graph,
    MFMA edge chain, the iteration replayed from a hipGraph; 6 calls of 64 structures."""
    from test_generator_gpu import _pkg
    import warnings
    P = _pkg()
    g = load_golden("dist_egnn_rc.npz")
    noise_kw = cases.noise_ns(100, **cases.LIN)
    sampling_kw = cases.sampling_ns(64, 1, M=2, one=False, greedy=False, cell=[10.86] * 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**noise_kw)
        spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=777, use_hip_graph=True)
    inner = nets.load_fixture_weights(nets.egnn_net(1, "radial_cutoff", 7.5), load_golden("traj_egnn_rc.npz"))
    net = nets.ScaledScore(inner, float(g["score_factor"])).to(cuda)
    net.edge_chain_precision = precision
    gen = P["Langevin"](npar, spar, net)

    def draw():
        calls = []
        with torch.no_grad():
            for _ in range(len(g["seeds"])):
                out = gen.sample(int(g["batch"]), cuda)
                assert (out.A == 0).all()
                calls.append(out.X.cpu().numpy())
        return calls

    assert judge_confirmed(g, draw) == []
    assert gen.f16_range_fallbacks == 0
    assert all(layer._chain[1] is not None and layer._chain[1].precision == precision for layer in inner.egnn.graph_layers)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,greedy_one", [("dist_egnn_types.npz", False), ("dist_egnn_types_greedy.npz", True)])
def test_two_atom_types_graph_loop_samples_the_reference_distribution(cuda, fixture, greedy_one):
    """Two atom types (configs[3]'s cell) in the fast mode: the atom-type update kernel with in-kernel Philox uniforms (Gumbel
    draws; or configs[3]'s greedy + one-transition settings) inside the captured iteration, around the small two-type EGNN with
    score x 100 and logits x 10 on both sides.  Held to the reference: the position scalars, the pair distances by equal /
    different type, the nearest atom of the own type, and the fraction of atoms that end as type 0; 6 calls of 64 structures."""
    from test_generator_gpu import _pkg
    import warnings
    P = _pkg()
    g = load_golden(fixture)
    noise_kw = cases.noise_ns(100, **cases.LIN)
    sampling_kw = cases.sampling_ns(64, 2, M=2, one=greedy_one, greedy=greedy_one, cell=[11.084] * 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**noise_kw)
        spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=999, use_hip_graph=True)
    inner = nets.load_fixture_weights(nets.egnn_net(2, "radial_cutoff", 7.5), load_golden("net_egnn_rc.npz"))
    net = nets.ScaledScore(inner, float(g["score_factor"]), float(g["logit_factor"])).to(cuda)
    gen = P["Langevin"](npar, spar, net)

    def draw():
        calls = []
        with torch.no_grad():
            for _ in range(len(g["seeds"])):
                out = gen.sample(int(g["batch"]), cuda)
                assert (out.A < 2).all()
                calls.append((out.X.cpu().numpy(), out.A.cpu().numpy()))
        return calls

    assert judge_confirmed(g, draw) == []
    assert gen.f16_range_fallbacks == 0


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [True, False])
def test_repaint_graph_loop_samples_the_reference_distribution(cuda, use_graph):
    """ConstrainedLangevinGenerator (repaint: BASELINE configs[4]'s algorithm) in the fast mode -- device Philox, the known
    rows noised and written by the HIP repaint kernel inside the captured iteration -- around the small radial-cutoff EGNN
    (score x 100), 32 of 64 atoms pinned at diamond sites: the distribution of the 32 FREE atoms (and of their distance to the
    nearest pinned site) against the reference's ConstrainedLangevinGenerator, 6 calls of 64 structures."""
    from test_generator_gpu import _pkg
    import warnings
    P = _pkg()
    g = load_golden("dist_egnn_repaint.npz")
    sites = torch.from_numpy(g["pinned_sites"])
    K = sites.shape[0]
    noise_kw = cases.noise_ns(100, **cases.LIN)
    sampling_kw = cases.sampling_ns(64, 1, M=2, one=False, greedy=False, cell=[10.86] * 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**noise_kw)
        spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=888, use_hip_graph=use_graph)
    inner = nets.load_fixture_weights(nets.egnn_net(1, "radial_cutoff", 7.5), load_golden("traj_egnn_rc.npz"))
    net = nets.ScaledScore(inner, float(g["score_factor"])).to(cuda)
    constraint = P["Constraint"](elements=["Si"], constrained_relative_coordinates=sites.clone(),
                                 constrained_atom_types=torch.zeros(K, dtype=torch.long))
    gen = P["Constrained"](npar, spar, net, constraint)

    def draw():
        calls = []
        with torch.no_grad():
            for _ in range(len(g["seeds"])):
                out = gen.sample(int(g["batch"]), cuda)
                assert (out.A == 0).all()
                x = out.X.cpu()
                u = x[:, :K] - sites[None]                  # (the last correctors move the known rows off their sites a little,
                assert float((u - u.round()).abs().max()) < 0.02          # as in the reference: the repaint is in the predictor)
                calls.append(x.numpy())
        return calls

    assert judge_confirmed(g, draw) == []
    assert gen.f16_range_fallbacks == 0


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_production_egnn_graph_loop_samples_the_reference_distribution(cuda, precision):
    """The network BASELINE configs[2] is quoted on (EGNN 4 x 256 x 4, rc 7.5, formula weights: the benchmarked kernel
    instantiation, egnn_edge_chain_kernel<256, ...>) over whole trajectories of T = 100 of configs[2]'s schedule, M = 2, N = 64,
    coordinate score x 150 on both sides; device Philox, hipGraph loop; 6 calls of 16 structures against the reference's 6 x 16."""
    from test_generator_gpu import _pkg
    import warnings
    P = _pkg()
    g = load_golden("dist_egnn_c3_wide.npz")
    noise_kw = cases.noise_ns(100, **cases.LIN)
    sampling_kw = cases.sampling_ns(64, 1, M=2, one=False, greedy=False, cell=[10.86] * 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**noise_kw)
        spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=4242, use_hip_graph=True)
    inner = nets.egnn_c3_net(1)
    net = nets.ScaledScore(inner, float(g["score_factor"])).to(cuda)
    net.edge_chain_precision = precision
    gen = P["Langevin"](npar, spar, net)

    def draw():
        calls = []
        with torch.no_grad():
            for _ in range(len(g["seeds"])):
                out = gen.sample(int(g["batch"]), cuda)
                assert (out.A == 0).all()
                calls.append(out.X.cpu().numpy())
        return calls

    assert judge_confirmed(g, draw) == []
    assert gen.f16_range_fallbacks == 0
    assert all(layer._chain[1] is not None and layer._chain[1].precision == precision for layer in inner.egnn.graph_layers)


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [True, False])
def test_update_kernels_sample_the_analytic_target(cuda, use_graph):
    """The per-step HIP path (schedule tables, fused predictor / corrector update with in-kernel Philox draws, the loop replayed
    from a hipGraph or launched eagerly) around the analytic score network as a PyTorch plugin: 24 calls of 1024 structures
    against the reference's 48 x 1024.  The case with power: the criterion rejects the reference's own runs with the score
    scaled by 0.97, and the final width depends on every factor of the update kernels (g^2, epsilon, sqrt(2 epsilon), sigma)."""
    from test_generator_gpu import _pkg
    import warnings
    P = _pkg()
    g = load_golden("dist_analytic.npz")
    noise_kw, sampling_kw, net = analytic_case(g)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**noise_kw)
        spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=1717, use_hip_graph=use_graph)
    gen = P["Langevin"](npar, spar, net.to(cuda))

    def draw():
        calls = []
        with torch.no_grad():
            for _ in range(24):
                out = gen.sample(int(g["batch"]), cuda)
                assert (out.A == 0).all()
                calls.append(out.X.cpu().numpy())
        return calls

    assert judge_confirmed(g, draw) == []
