"""Reference-made DISTRIBUTIONS of whole sampling jobs (container-only: imports /root/reference through make_golden's stubs).

    PYTHONPATH=/root/reference/src python tests/golden/make_distributions.py [analytic|mlp|mlp_well|egnn|egnn_c3_wide|egnn_repaint|egnn_types|egnn_types_greedy]

Runs the REFERENCE's LangevinGenerator for complete trajectories where that is cheap, several seeds each, and stores summary
statistics only -- quantile tables of the pooled final structures' scalars (tests/distribution_stats.py), the reference-vs-
reference two-sample KS distances across seeds (the calibration of what "the same distribution" looks like at this sample size),
and the KS distance of deliberately WRONG samplers to the pool (the power of the check):

  dist_mlp_c2.npz   BASELINE configs[1]: the MLP template (weights = tests/golden/net_mlp_c1.npz: `_mlp(8, 1)`), N = 8, T = 1000,
                    sigma 1e-4 .. 0.25 exponential, M = 1, greedy + one-transition defaults, 1024 structures per seed
                    (src/.../generators/langevin_generator.py:27-831 through src/.../sampling/diffusion_sampling.py:16-73)
  dist_mlp_well.npz the same job with the MLP template turned into a periodic well by its weights (tests/cases.py::
                    periodic_well_mlp_state): the case with power for the MLP samplers, 1024 structures x 16 seeds
  dist_analytic.npz the reference's AnalyticalScoreNetwork (exact score of Gaussians of width 0.05 around the diamond sites of
                    Si 1x1x1), T = 200, sigma 1e-4 .. 0.25 exponential, M = 1, 1024 structures x 48 seeds: the case with POWER
  dist_egnn_rc.npz  a small radial-cutoff EGNN (hidden 32, 2 graph layers; weights = tests/golden/traj_egnn_rc.npz), N = 64,
                    cell 10.86, T = 100 of configs[2]'s linear schedule, M = 2, 64 structures per seed; its coordinate score x 100
                    (see egnn_rc below: with the bare random-init network the check would have no power)
  dist_egnn_repaint.npz  ConstrainedLangevinGenerator (repaint, configs[4]'s algorithm) around the same small EGNN: 32 of the 64 atoms
                    pinned at diamond sites, the statistics of the 32 free ones
  dist_egnn_types.npz, dist_egnn_types_greedy.npz  two atom types (configs[3]'s cell): the small EGNN with num_atom_types = 2, types
                    drawn with Gumbel noise / configs[3]'s greedy + one-transition settings; logits x 10
  dist_egnn_c3_wide.npz  the PRODUCTION network of configs[2] (EGNN 4 x 256 x 4, rc 7.5, formula weights), same schedule and
                    T = 100, 16 structures x 6 seeds; coordinate score x 150

No structure, draw or source text is stored: the tables hold 2049 quantiles per scalar.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # (tests/cases.py imports the product's torch modules)
import make_golden as G  # noqa: E402  (installs the stubs, imports the reference)
import distribution_stats as DS  # noqa: E402

EGNN_SCORE_FACTOR = 100.0
LIN = dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8)


class ScaledScore(torch.nn.Module):
    """A deliberately wrong network for the power probes: the wrapped network's coordinate score times a factor (and its
    atom-type logits times `logit_factor`; the MASK class's -inf stays -inf)."""

    def __init__(self, net, factor, logit_factor=1.0):
        super().__init__()
        self.net, self.factor, self.logit_factor = net, factor, logit_factor

    def forward(self, batch, conditional=False):
        out = self.net(batch, conditional=conditional)
        A = out.A if self.logit_factor == 1.0 else torch.where(torch.isinf(out.A), out.A, out.A * self.logit_factor)
        return G.AXL(A=A, X=out.X * self.factor, L=out.L)


def run(make, seeds, batch, per_atom, name, probes, extra=None, sites=None, probe_calls=1, pinned=None, with_types=False):
    per_seed, fractions = [], []

    def stats(axl):
        return DS.statistics(axl.X.numpy(), per_atom=per_atom, sites=sites, pinned=pinned,
                             types=axl.A.numpy() if with_types else None)

    for seed in seeds:
        gen = make()
        torch.manual_seed(seed)
        t0 = time.perf_counter()
        with torch.no_grad():
            axl = gen.sample(batch, torch.device("cpu"))
        assert (axl.A != gen.num_classes - 1).all()
        per_seed.append(stats(axl))
        fractions.append(float((axl.A == 0).double().mean()))
        print(f"{name}: seed {seed} done in {time.perf_counter() - t0:.1f} s", flush=True)
    keys = list(per_seed[0])
    out = {"seeds": np.array(seeds), "batch": np.array(batch), "scalars": np.array(keys)}
    for key in keys:
        pool = np.concatenate([s[key] for s in per_seed])
        # (2049 knots for the pooled scalars, 513 for the per-atom marginals, whose samples are B values per seed; binary32)
        out[f"table/{key}"] = DS.quantile_table(pool, knots=513 if key.startswith("atom") else 2049).astype(np.float32)
        # calibration: every seed against the pool of the OTHER seeds, and every pair of seeds
        out[f"leave_one_out/{key}"] = np.array([DS.ks_two_sample(s[key], np.concatenate(
            [t[key] for j, t in enumerate(per_seed) if j != i])) for i, s in enumerate(per_seed)])
        out[f"pairwise/{key}"] = np.array([DS.ks_two_sample(per_seed[i][key], per_seed[j][key])
                                           for i in range(len(seeds)) for j in range(i + 1, len(seeds))])
        # the table's own resolution: a seed against the table vs against the exact pool (all seeds)
        out[f"seed_vs_table/{key}"] = np.array([DS.ks_to_table(s[key], out[f"table/{key}"].astype(np.float64)) for s in per_seed])
        # pooled halves against each other (20 random splits of the seeds): what two POOLED samples of the same
        # distribution look like -- the calibration for a pooled sample of the product against the table
        rng = np.random.default_rng(2024)
        halves = []
        for _ in range(20):
            order = rng.permutation(len(seeds))
            a = np.concatenate([per_seed[i][key] for i in order[: len(seeds) // 2]])
            b = np.concatenate([per_seed[i][key] for i in order[len(seeds) // 2:]])
            halves.append(DS.ks_two_sample(a, b))
        out[f"half_split/{key}"] = np.array(halves)
    out["probe_calls"] = np.array(probe_calls)
    if with_types:          # the fraction of atoms that end as type 0, per seed (one number per call: held by its spread, not by KS)
        out["type_fraction/per_seed"] = np.array(fractions)
        print(f"{name}: fraction of type 0 per seed {np.round(fractions, 4)}", flush=True)
    for probe, make_wrong in probes.items():
        calls = []
        for c in range(probe_calls):          # the wrong sampler's calls: the first alone, and all of them pooled
            gen = make_wrong()
            torch.manual_seed(seeds[0] + 1000 * c)
            with torch.no_grad():
                axl = gen.sample(batch, torch.device("cpu"))
            calls.append(stats(axl))
            if with_types and c == 0:
                out[f"probe_type_fraction/{probe}"] = np.array(float((axl.A == 0).double().mean()))
        for key in keys:
            table = out[f"table/{key}"].astype(np.float64)
            out[f"probe/{probe}/{key}"] = np.array(DS.ks_to_table(calls[0][key], table))
            if probe_calls > 1:
                out[f"probe_pooled/{probe}/{key}"] = np.array(DS.ks_to_table(np.concatenate([c[key] for c in calls]), table))
        print(f"{name}: probe {probe}: " + ", ".join(f"{k} {float(out[f'probe/{probe}/{k}']):.3f}" for k in keys[:8]) +
              (f", type fraction {float(out[f'probe_type_fraction/{probe}']):.4f}" if with_types else "") +
              (" | pooled: " + ", ".join(f"{k} {float(out[f'probe_pooled/{probe}/{k}']):.4f}" for k in keys[:5]) if probe_calls > 1 else ""),
              flush=True)
    for key in keys[:8]:
        print(f"{name}: {key}: leave-one-out {out[f'leave_one_out/{key}'].round(4)}  pairwise max {out[f'pairwise/{key}'].max():.4f}")
    out.update(extra or {})
    G.save(name + ".npz", **out)


def mlp_c2():
    kw = dict(T=1000, N=8, num_atom_types=1, M=1, noise_kw=dict(sigma_min=1e-4, sigma_max=0.25))

    def make(net=None, **over):
        return G.make_generator(net=net or G._mlp(8, 1), **dict(kw, **over))[0]
    probes = {
        "zero_score": lambda: make(net=ScaledScore(G._mlp(8, 1), 0.0)),
        "score_x0.9": lambda: make(net=ScaledScore(G._mlp(8, 1), 0.9)),
        "no_corrector": lambda: make(M=0),
        "sigma_max_0.2": lambda: make(noise_kw=dict(sigma_min=1e-4, sigma_max=0.2)),
    }
    run(make, seeds=list(range(11, 27)), batch=1024, per_atom=True, name="dist_mlp_c2", probes=probes)


WELL = dict(amplitude=0.3, offset=24.0)


def mlp_well():
    """BASELINE configs[1]'s job (N = 8, T = 1000, sigma 1e-4 .. 0.25 exponential, M = 1, 1024 structures per seed) with the MLP
    template made a KNOWN function by its weights (tests/cases.py::periodic_well_mlp_state: out.X = -0.3 sin(2 pi (x - site)),
    the diamond sites of Si 1x1x1): where the random-init template's chaotic map leaves the criterion blind to a 10 % error of
    the score (dist_mlp_c2), this one contracts every atom onto its site with a width set by the last ~150 corrector steps --
    the case with power for samplers that take an MLP (the persistent fused kernel).  16 seeds; the wrong samplers 8 calls each."""
    sys.path.insert(0, os.path.dirname(HERE))
    from cases import diamond_sites, periodic_well_mlp_state
    sites = diamond_sites(1)
    kw = dict(T=1000, N=8, num_atom_types=1, M=1, noise_kw=dict(sigma_min=1e-4, sigma_max=0.25))

    def network(factor=1.0):
        net = G._mlp(8, 1)
        state = periodic_well_mlp_state(sites.numpy(), factor=factor, reference_shapes={k: v.shape for k, v in net.state_dict().items()},
                                        **WELL)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        return net

    def make(factor=1.0, **over):
        return G.make_generator(net=network(factor), **dict(kw, **over))[0]
    probes = {
        "zero_score": lambda: make(factor=0.0),
        "score_x0.9": lambda: make(factor=0.9),
        "score_x0.97": lambda: make(factor=0.97),
        "no_corrector": lambda: make(M=0),
        "sigma_min_1e-3": lambda: make(noise_kw=dict(sigma_min=1e-3, sigma_max=0.25)),
    }
    g = torch.Generator().manual_seed(5050)
    B = 6
    x = torch.rand(B, 8, 3, generator=g)
    batch = {G.NOISY_AXL_COMPOSITION: G.AXL(A=torch.zeros(B, 8, dtype=torch.long), X=x,
                                            L=torch.tensor([5.43, 5.43, 5.43, 0, 0, 0.0]).repeat(B, 1)),
             G.TIME: torch.rand(B, 1, generator=g), G.NOISE: torch.rand(B, 1, generator=g) * 0.25,
             G.CARTESIAN_FORCES: torch.zeros(B, 8, 3)}
    with torch.no_grad():
        out = network()(batch, conditional=False)
    extra = {"forward/X": G._np(x), "forward/time": G._np(batch[G.TIME]), "forward/sigma": G._np(batch[G.NOISE]),
             "forward/out_X": G._np(out.X), "forward/out_A": G._np(out.A), "sites": G._np(sites),
             "well": np.array([WELL["amplitude"], WELL["offset"]], dtype=np.float64)}
    run(make, seeds=list(range(201, 217)), batch=1024, per_atom=False, name="dist_mlp_well", probes=probes, extra=extra,
        sites=G._np(sites), probe_calls=8)


def egnn_rc():
    kw = dict(T=100, N=64, num_atom_types=1, M=2, one=False, greedy=False, cell=[10.86] * 3, noise_kw=LIN)

    # A randomly initialised EGNN of this size returns sigma-normalised scores of 3e-3 rms: the reverse process then ends in
    # the uniform distribution whatever the sampler does with the score (measured: a run with the score zeroed is inside the
    # seed-to-seed spread), and a distributional check would have no power.  The network under test is therefore the EGNN with
    # its coordinate score multiplied by EGNN_SCORE_FACTOR (rms 0.3: the drift over the trajectory is ~0.1 of the cell,
    # comparable with the nearest-neighbour distance) -- a plugin wrapped around the EGNN on both sides.
    def make(net=None, factor=EGNN_SCORE_FACTOR, **over):
        return G.make_generator(net=net or ScaledScore(G._egnn(1, "radial_cutoff", 7.5), factor), **dict(kw, **over))[0]
    probes = {
        "zero_score": lambda: make(factor=0.0),
        "score_x0.5": lambda: make(factor=0.5 * EGNN_SCORE_FACTOR),
        "score_x0.9": lambda: make(factor=0.9 * EGNN_SCORE_FACTOR),
        "no_corrector": lambda: make(M=0),
    }
    run(make, seeds=[21, 22, 23, 24, 25, 26], batch=64, per_atom=False, name="dist_egnn_rc", probes=probes,
        extra={"score_factor": np.array(EGNN_SCORE_FACTOR)})


def egnn_repaint():
    """The reference's ConstrainedLangevinGenerator (src/.../generators/constrained_langevin_generator.py:94-163: the repaint
    algorithm BASELINE configs[4] samples with) over whole trajectories: the small radial-cutoff EGNN of egnn_rc (score x 100),
    N = 64, cell 10.86, T = 100, M = 2, the first K = 32 atoms pinned at the first 32 diamond sites of Si 2x2x2
    (constrained_indices = arange), 64 structures per seed.  The statistics are those of the 32 FREE atoms (the pinned rows end
    at their sites by construction), `to_pinned` included.  The probe `no_repaint` is the unconstrained generator: what the
    free atoms' distribution would be without the conditioning."""
    from cases import diamond_sites
    K = 32
    sites = diamond_sites(2)[:K].clone()
    kw = dict(T=100, N=64, num_atom_types=1, M=2, one=False, greedy=False, cell=[10.86] * 3, noise_kw=LIN)

    def make(factor=EGNN_SCORE_FACTOR, constrained=True, **over):
        constraint = G.SamplingConstraint(elements=["Si"], constrained_relative_coordinates=sites.clone(),
                                          constrained_atom_types=torch.zeros(K, dtype=torch.long)) if constrained else None
        gen = G.make_generator(net=ScaledScore(G._egnn(1, "radial_cutoff", 7.5), factor), constraint=constraint,
                               **dict(kw, **over))[0]
        if constrained:
            assert torch.equal(gen.constraint_indices, torch.arange(K))
        return gen
    probes = {
        "zero_score": lambda: make(factor=0.0),
        "score_x0.5": lambda: make(factor=0.5 * EGNN_SCORE_FACTOR),
        "no_corrector": lambda: make(M=0),
        "no_repaint": lambda: make(constrained=False),
    }
    run(make, seeds=[41, 42, 43, 44, 45, 46], batch=64, per_atom=False, name="dist_egnn_repaint", probes=probes,
        extra={"score_factor": np.array(EGNN_SCORE_FACTOR), "pinned_sites": G._np(sites)}, pinned=G._np(sites))


LOGIT_FACTOR = 10.0


def egnn_two_types(greedy_one):
    """Two atom types (BASELINE configs[3]'s SiGe cell, 11.084): the small radial-cutoff EGNN with num_atom_types = 2 (weights =
    tests/golden/net_egnn_rc.npz), N = 64, T = 100 of configs[2]'s schedule, M = 2, 64 structures per seed; coordinate score
    x 100 and atom-type LOGITS x 10 on both sides (the random-init network's logits differ by 0.2: every class probability
    would be 0.5 +- 0.05 whatever the sampler does with them).  Two settings of the atom-type update
    (src/.../generators/langevin_generator.py:247-439): greedy_one = False -- types drawn with Gumbel noise, any number of
    transitions per step (the device-Philox uniforms of the fast mode at work); greedy_one = True -- configs[3]'s own settings
    (atom_type_greedy_sampling and one_atom_type_transition_per_step on).  Stored besides the position scalars: pair distances
    by equal / different type, the nearest atom of the own type, and the fraction of atoms that end as type 0 per seed."""
    kw = dict(T=100, N=64, num_atom_types=2, M=2, one=greedy_one, greedy=greedy_one, cell=[11.084] * 3, noise_kw=LIN)

    def make(factor=EGNN_SCORE_FACTOR, logit_factor=LOGIT_FACTOR, **over):
        net = ScaledScore(G._egnn(2, "radial_cutoff", 7.5), factor, logit_factor)
        return G.make_generator(net=net, **dict(kw, **over))[0]
    probes = {
        "zero_score": lambda: make(factor=0.0),
        "uniform_types": lambda: make(logit_factor=0.0),
        "logits_x0.5": lambda: make(logit_factor=0.5 * LOGIT_FACTOR),
        "other_type_update": lambda: make(one=not greedy_one, greedy=not greedy_one),
    }
    name = "dist_egnn_types_greedy" if greedy_one else "dist_egnn_types"
    # (12 seeds under the greedy settings: its first six happened to lie within 0.0018 .. 0.0021 of each other on `pair` -- the
    # same scalar spreads over 0.0019 .. 0.0035 in dist_egnn_rc / dist_egnn_types -- and six draws do not show the tail)
    seeds = list(range(51, 63)) if greedy_one else [51, 52, 53, 54, 55, 56]
    run(make, seeds=seeds, batch=64, per_atom=False, name=name, probes=probes, with_types=True,
        extra={"score_factor": np.array(EGNN_SCORE_FACTOR), "logit_factor": np.array(LOGIT_FACTOR)})


C3_WIDE_SCORE_FACTOR = 150.0


def egnn_c3_wide():
    """The network BASELINE configs[2] is quoted on -- the reference's EGNN 4 x 256 x 4, radial cutoff 7.5, weights by
    tests/formula_weights.py as in net_egnn_c3.npz -- over whole (shortened) trajectories: N = 64, cell 10.86, T = 100 of
    configs[2]'s linear schedule, M = 2, 16 structures per seed (1.4 s per iteration on this container's 8 cores).  The formula
    network's sigma-normalised scores are 2e-3 rms, so, as in egnn_rc, the coordinate score is multiplied by a constant
    (C3_WIDE_SCORE_FACTOR: rms 0.3) by a plugin on both sides.  Under that score the 64 atoms of a structure end in a few tight
    clusters: the translation-invariant scalars (pair, nn) then separate samplers sharply (zeroed score: KS 0.99; halved: 0.49;
    seed to seed: 0.035), while the per-axis coordinates only say where a structure's clusters landed (16 structures per call:
    seed-to-seed KS 0.08 .. 0.33, the zero-score probe inside it) -- `judged` names the scalars the criterion uses."""
    kw = dict(T=100, N=64, num_atom_types=1, M=2, one=False, greedy=False, cell=[10.86] * 3, noise_kw=LIN)
    inner = G._egnn_c3(1)

    def make(factor=C3_WIDE_SCORE_FACTOR, **over):
        return G.make_generator(net=ScaledScore(inner, factor), **dict(kw, **over))[0]
    probes = {
        "zero_score": lambda: make(factor=0.0),
        "score_x0.5": lambda: make(factor=0.5 * C3_WIDE_SCORE_FACTOR),
        "score_x0.9": lambda: make(factor=0.9 * C3_WIDE_SCORE_FACTOR),
        "no_corrector": lambda: make(M=0),
    }
    run(make, seeds=[31, 32, 33, 34, 35, 36], batch=16, per_atom=False, name="dist_egnn_c3_wide", probes=probes,
        extra={"score_factor": np.array(C3_WIDE_SCORE_FACTOR), "judged": np.array(["pair", "nn"])})


ANALYTIC = dict(number_of_atoms=8, kmax=4, sigma_d=0.05, T=200, sigma_min=1e-4, sigma_max=0.25)


def analytic():
    """A case where the score MATTERS by construction: the reference's AnalyticalScoreNetwork
    (src/.../models/score_networks/analytical_score_network.py:63-298) -- the exact score of independent wrapped Gaussians of
    width sigma_d around the eight diamond sites of Si 1x1x1 -- under the reference's LangevinGenerator (T = 200, sigma
    1e-4 .. 0.25 exponential, M = 1): the sampler must CONTRACT the uniform initial cloud onto the sites, and the final width
    depends on every factor of the update (g^2, epsilon, the score's weight).  1024 structures x 48 seeds.  Besides the
    statistics of the other cases, `disp` = the displacement of every coordinate from its site (wrapped to [-1/2, 1/2)).
    Also stored: a forward of the reference's network on random inputs, to pin the tests' own restatement of it."""
    from diffusion_for_multi_scale_molecular_dynamics.models.score_networks.analytical_score_network import (
        AnalyticalScoreNetwork, AnalyticalScoreNetworkParameters)
    sys.path.insert(0, os.path.dirname(HERE))
    from cases import diamond_sites
    sites = diamond_sites(1)
    c = ANALYTIC

    def network():
        return AnalyticalScoreNetwork(AnalyticalScoreNetworkParameters(
            number_of_atoms=c["number_of_atoms"], kmax=c["kmax"], sigma_d=c["sigma_d"], spatial_dimension=3, num_atom_types=1,
            equilibrium_relative_coordinates=sites.tolist()))

    kw = dict(T=c["T"], N=c["number_of_atoms"], num_atom_types=1, M=1,
              noise_kw=dict(sigma_min=c["sigma_min"], sigma_max=c["sigma_max"]))

    def make(net=None, **over):
        return G.make_generator(net=net or network(), **dict(kw, **over))[0]
    probes = {
        "zero_score": lambda: make(net=ScaledScore(network(), 0.0)),
        "score_x0.9": lambda: make(net=ScaledScore(network(), 0.9)),
        "score_x0.97": lambda: make(net=ScaledScore(network(), 0.97)),
        "no_corrector": lambda: make(M=0),
        "sigma_max_0.2": lambda: make(noise_kw=dict(sigma_min=c["sigma_min"], sigma_max=0.2)),
        "sigma_min_1e-2": lambda: make(noise_kw=dict(sigma_min=1e-2, sigma_max=c["sigma_max"])),
    }
    g = torch.Generator().manual_seed(4040)
    B = 6
    x = torch.rand(B, c["number_of_atoms"], 3, generator=g)
    sig = torch.tensor([1e-4, 1e-3, 1e-2, 0.05, 0.15, 0.25]).reshape(B, 1)
    batch = {G.NOISY_AXL_COMPOSITION: G.AXL(A=torch.zeros(B, c["number_of_atoms"], dtype=torch.long), X=x,
                                            L=torch.tensor([5.43, 5.43, 5.43, 0, 0, 0.0]).repeat(B, 1)),
             G.TIME: torch.rand(B, 1, generator=g), G.NOISE: sig, G.CARTESIAN_FORCES: torch.zeros(B, c["number_of_atoms"], 3)}
    with torch.no_grad():
        out = network()(batch, conditional=False)
    extra = {"forward/X": G._np(x), "forward/sigma": G._np(sig), "forward/out_X": G._np(out.X), "forward/out_A": G._np(out.A),
             "sites": G._np(sites), "settings": np.array([c[k] for k in ("number_of_atoms", "kmax", "sigma_d", "T", "sigma_min", "sigma_max")],
                                                          dtype=np.float64)}
    run(make, seeds=list(range(101, 149)), batch=1024, per_atom=False, name="dist_analytic", probes=probes, extra=extra,
        sites=G._np(sites), probe_calls=24)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "analytic"):
        analytic()
    if which in ("all", "mlp"):
        mlp_c2()
    if which in ("all", "mlp_well"):
        mlp_well()
    if which in ("all", "egnn"):
        egnn_rc()
    if which in ("all", "egnn_c3_wide"):
        egnn_c3_wide()
    if which in ("all", "egnn_repaint"):
        egnn_repaint()
    if which in ("all", "egnn_types"):
        egnn_two_types(False)
    if which in ("all", "egnn_types_greedy"):
        egnn_two_types(True)
