"""Reference-made DISTRIBUTIONS of whole sampling jobs (container-only: imports /root/reference through make_golden's stubs).

    PYTHONPATH=/root/reference/src python tests/golden/make_distributions.py [mlp|egnn]

Runs the REFERENCE's LangevinGenerator for complete trajectories where that is cheap, several seeds each, and stores summary
statistics only -- quantile tables of the pooled final structures' scalars (tests/distribution_stats.py), the reference-vs-
reference two-sample KS distances across seeds (the calibration of what "the same distribution" looks like at this sample size),
and the KS distance of deliberately WRONG samplers to the pool (the power of the check):

  dist_mlp_c2.npz   BASELINE configs[1]: the MLP template (weights = tests/golden/net_mlp_c1.npz: `_mlp(8, 1)`), N = 8, T = 1000,
                    sigma 1e-4 .. 0.25 exponential, M = 1, greedy + one-transition defaults, 1024 structures per seed
                    (src/.../generators/langevin_generator.py:27-831 through src/.../sampling/diffusion_sampling.py:16-73)
  dist_egnn_rc.npz  a small radial-cutoff EGNN (hidden 32, 2 graph layers; weights = tests/golden/traj_egnn_rc.npz), N = 64,
                    cell 10.86, T = 100 of configs[2]'s linear schedule, M = 2, 64 structures per seed; its coordinate score x 100
                    (see egnn_rc below: with the bare random-init network the check would have no power)

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
import make_golden as G  # noqa: E402  (installs the stubs, imports the reference)
import distribution_stats as DS  # noqa: E402

EGNN_SCORE_FACTOR = 100.0
LIN = dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8)


class ScaledScore(torch.nn.Module):
    """A deliberately wrong network for the power probes: the wrapped network's coordinate score times a factor."""

    def __init__(self, net, factor):
        super().__init__()
        self.net, self.factor = net, factor

    def forward(self, batch, conditional=False):
        out = self.net(batch, conditional=conditional)
        return G.AXL(A=out.A, X=out.X * self.factor, L=out.L)


def run(make, seeds, batch, per_atom, name, probes, extra=None):
    per_seed = []
    for seed in seeds:
        gen = make()
        torch.manual_seed(seed)
        t0 = time.perf_counter()
        with torch.no_grad():
            axl = gen.sample(batch, torch.device("cpu"))
        assert (axl.A != gen.num_classes - 1).all()
        per_seed.append(DS.statistics(axl.X.numpy(), per_atom=per_atom))
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
    for probe, make_wrong in probes.items():
        gen = make_wrong()
        torch.manual_seed(seeds[0])
        with torch.no_grad():
            axl = gen.sample(batch, torch.device("cpu"))
        stats = DS.statistics(axl.X.numpy(), per_atom=per_atom)
        for key in keys:
            out[f"probe/{probe}/{key}"] = np.array(DS.ks_to_table(stats[key], out[f"table/{key}"].astype(np.float64)))
        print(f"{name}: probe {probe}: " + ", ".join(f"{k} {float(out[f'probe/{probe}/{k}']):.3f}" for k in keys[:5]), flush=True)
    for key in keys[:5]:
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


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "mlp"):
        mlp_c2()
    if which in ("all", "egnn"):
        egnn_rc()
