"""KS distances of the product's fast modes to the reference-made statistics, next to the limits the tests apply
(tests/test_distribution.py; fixtures tests/golden/dist_*.npz).  Run on the GPU box:

    python tools/distribution_report.py > gpurun_out/distribution_report.txt
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import nets  # noqa: E402
import test_distribution as T  # noqa: E402
from conftest import load_golden  # noqa: E402
from test_generator_gpu import _pkg  # noqa: E402

cuda = torch.device("cuda:0")
P = _pkg()


def summary(title, g, calls):
    rows = T.measure(g, calls)
    first = calls[0][0] if isinstance(calls[0], tuple) else calls[0]
    print(f"== {title}: {len(calls)} calls of {first.shape[0]} structures")
    groups = {}
    for what, d, limit in rows:
        key = what.split(": ", 1)[1] if what.startswith("call") else what
        groups.setdefault(key, []).append((d, limit))
    for key, vals in groups.items():
        d = np.array([v[0] for v in vals])
        print(f"   {key:38s} largest {d.max():.4f}  mean {d.mean():.4f}  limit {vals[0][1]:.4f}  "
              f"(reference seeds, leave-one-out: {reference_spread(g, key)})")
    print("   verdict:", "passes" if not T.judge(g, calls) else T.judge(g, calls))


def reference_spread(g, key):
    k = key.replace("pooled: ", "")
    if f"leave_one_out/{k}" in g.files:
        v = g[f"half_split/{k}"] if key.startswith("pooled") else g[f"leave_one_out/{k}"]
        return f"largest {v.max():.4f} mean {v.mean():.4f}"
    return "-"


with torch.no_grad(), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    g = load_golden("dist_mlp_c2.npz")
    noise_kw, sampling_kw = T.mlp_c2_parameters()
    net = nets.load_fixture_weights(nets.mlp_net(8, 1), load_golden("net_mlp_c1.npz")).to(cuda)
    gen = P["Langevin"](P["Noise"](**noise_kw), P["Sampling"](**sampling_kw, rng_mode="device", seed=9090, fused_score_network=True), net)
    summary("C2 MLP, persistent fused kernel (device Philox)", g,
            [gen.sample(int(g["batch"]), cuda).X.cpu().numpy() for _ in range(len(g["seeds"]))])
    for probe in ("zero_score", "score_x0.9", "no_corrector", "sigma_max_0.2"):
        atoms = [str(k) for k in g["scalars"] if str(k).startswith("atom")]
        d = np.array([float(g[f"probe/{probe}/{k}"]) for k in atoms])
        print(f"   reference-side wrong sampler {probe:14s}: " + ", ".join(f"{k} {float(g[f'probe/{probe}/{k}']):.4f}" for k in T.POOLED) +
              f", largest / mean per-atom marginal {d.max():.4f} / {d.mean():.4f}  -> {'rejected' if T.probe_fails(g, probe) else 'not seen'}")

    g = load_golden("dist_analytic.npz")
    noise_kw, sampling_kw, net = T.analytic_case(g)
    for use_graph in (True, False):
        gen = P["Langevin"](P["Noise"](**noise_kw), P["Sampling"](**sampling_kw, rng_mode="device", seed=1717, use_hip_graph=use_graph),
                            net.to(cuda))
        summary(f"analytic Gaussian-well score (plugin), N = 8, T = 200, per-step HIP kernels, {'hipGraph' if use_graph else 'eager'}", g,
                [gen.sample(int(g["batch"]), cuda).X.cpu().numpy() for _ in range(24)])
    for probe in ("zero_score", "score_x0.9", "score_x0.97", "no_corrector", "sigma_max_0.2", "sigma_min_1e-2"):
        keys = T.scalars_of(g)
        print(f"   reference-side wrong sampler {probe:14s}: first call " + ", ".join(f"{k} {float(g[f'probe/{probe}/{k}']):.4f}" for k in keys) +
              f" | {int(g['probe_calls'])} calls pooled " + ", ".join(f"{k} {float(g[f'probe_pooled/{probe}/{k}']):.4f}" for k in keys) +
              f"  -> {'rejected' if T.probe_fails(g, probe) else 'not seen'}")

    g = load_golden("dist_egnn_rc.npz")
    for precision in ("f16x3", "f32"):
        inner = nets.load_fixture_weights(nets.egnn_net(1, "radial_cutoff", 7.5), load_golden("traj_egnn_rc.npz"))
        net = nets.ScaledScore(inner, float(g["score_factor"])).to(cuda)
        net.edge_chain_precision = precision
        gen = P["Langevin"](P["Noise"](**cases.noise_ns(100, **cases.LIN)),
                            P["Sampling"](**cases.sampling_ns(64, 1, M=2, one=False, greedy=False, cell=[10.86] * 3),
                                          rng_mode="device", seed=777, use_hip_graph=True), net)
        summary(f"EGNN hidden 32, N = 64, T = 100, score x {float(g['score_factor']):.0f}, hipGraph loop, edge chain {precision}", g,
                [gen.sample(int(g["batch"]), cuda).X.cpu().numpy() for _ in range(len(g["seeds"]))])
    for probe in ("zero_score", "score_x0.5", "score_x0.9", "no_corrector"):
        print(f"   reference-side wrong sampler {probe:14s}: " + ", ".join(f"{k} {float(g[f'probe/{probe}/{k}']):.4f}" for k in T.POOLED) +
              f"  -> {'rejected' if T.probe_fails(g, probe) else 'not seen'}")

    g = load_golden("dist_egnn_c3_wide.npz")
    for precision in ("f16x3", "f32"):
        net = nets.ScaledScore(nets.egnn_c3_net(1), float(g["score_factor"])).to(cuda)
        net.edge_chain_precision = precision
        gen = P["Langevin"](P["Noise"](**cases.noise_ns(100, **cases.LIN)),
                            P["Sampling"](**cases.sampling_ns(64, 1, M=2, one=False, greedy=False, cell=[10.86] * 3),
                                          rng_mode="device", seed=4242, use_hip_graph=True), net)
        summary(f"production EGNN 4 x 256 x 4 (configs[2]'s network), N = 64, T = 100, score x {float(g['score_factor']):.0f}, "
                f"hipGraph loop, edge chain {precision}", g,
                [gen.sample(int(g["batch"]), cuda).X.cpu().numpy() for _ in range(len(g["seeds"]))])
    for probe in ("zero_score", "score_x0.5", "score_x0.9", "no_corrector"):
        print(f"   reference-side wrong sampler {probe:14s}: " + ", ".join(f"{k} {float(g[f'probe/{probe}/{k}']):.4f}" for k in T.POOLED) +
              f"  (judged on {', '.join(T.scalars_of(g))})  -> {'rejected' if T.probe_fails(g, probe) else 'not seen'}")

    if os.path.exists(os.path.join(ROOT, "tests", "golden", "dist_egnn_repaint.npz")):
        g = load_golden("dist_egnn_repaint.npz")
        sites = torch.from_numpy(g["pinned_sites"])
        for use_graph in (True, False):
            inner = nets.load_fixture_weights(nets.egnn_net(1, "radial_cutoff", 7.5), load_golden("traj_egnn_rc.npz"))
            net = nets.ScaledScore(inner, float(g["score_factor"])).to(cuda)
            constraint = P["Constraint"](elements=["Si"], constrained_relative_coordinates=sites.clone(),
                                         constrained_atom_types=torch.zeros(sites.shape[0], dtype=torch.long))
            gen = P["Constrained"](P["Noise"](**cases.noise_ns(100, **cases.LIN)),
                                   P["Sampling"](**cases.sampling_ns(64, 1, M=2, one=False, greedy=False, cell=[10.86] * 3),
                                                 rng_mode="device", seed=888, use_hip_graph=use_graph), net, constraint)
            summary(f"repaint (ConstrainedLangevinGenerator), EGNN hidden 32, 32 of 64 atoms pinned, T = 100, "
                    f"{'hipGraph' if use_graph else 'eager'}: the 32 free atoms", g,
                    [gen.sample(int(g["batch"]), cuda).X.cpu().numpy() for _ in range(len(g["seeds"]))])
        for probe in ("zero_score", "score_x0.5", "no_corrector", "no_repaint"):
            print(f"   reference-side wrong sampler {probe:14s}: " + ", ".join(f"{k} {float(g[f'probe/{probe}/{k}']):.4f}" for k in T.scalars_of(g)) +
                  f"  -> {'rejected' if T.probe_fails(g, probe) else 'not seen'}")

    for fixture, greedy_one in (("dist_egnn_types.npz", False), ("dist_egnn_types_greedy.npz", True)):
        if not os.path.exists(os.path.join(ROOT, "tests", "golden", fixture)):
            continue
        g = load_golden(fixture)
        inner = nets.load_fixture_weights(nets.egnn_net(2, "radial_cutoff", 7.5), load_golden("net_egnn_rc.npz"))
        net = nets.ScaledScore(inner, float(g["score_factor"]), float(g["logit_factor"])).to(cuda)
        gen = P["Langevin"](P["Noise"](**cases.noise_ns(100, **cases.LIN)),
                            P["Sampling"](**cases.sampling_ns(64, 2, M=2, one=greedy_one, greedy=greedy_one, cell=[11.084] * 3),
                                          rng_mode="device", seed=999, use_hip_graph=True), net)
        calls = []
        for _ in range(len(g["seeds"])):
            out = gen.sample(int(g["batch"]), cuda)
            calls.append((out.X.cpu().numpy(), out.A.cpu().numpy()))
        summary(f"two atom types, EGNN hidden 32, N = 64, T = 100, logits x {float(g['logit_factor']):.0f}, "
                f"{'greedy + one transition per step' if greedy_one else 'Gumbel draws'}, hipGraph loop", g, calls)
        mean, limit = T.type_fraction_limits(g)
        print(f"   fraction of type 0: reference seeds {np.round(g['type_fraction/per_seed'], 4)} (mean {mean:.4f}, limit +- {limit:.4f}); "
              f"product calls {np.round([float((a == 0).mean()) for _, a in calls], 4)}")
        for probe in ("zero_score", "uniform_types", "logits_x0.5", "other_type_update"):
            print(f"   reference-side wrong sampler {probe:18s}: " + ", ".join(f"{k} {float(g[f'probe/{probe}/{k}']):.4f}" for k in T.scalars_of(g)) +
                  f", type fraction {float(g[f'probe_type_fraction/{probe}']):.4f}  -> {'rejected' if T.probe_fails(g, probe) else 'not seen'}")

    g = load_golden("dist_mlp_well.npz")
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
    noise_kw, sampling_kw, net = T.periodic_well_case(g)
    for variant, options in (("register-resident exact family (bench.py's C2 path)", 0), ("padded family", _hip.MLP_SAMPLE_PADDED_FAMILY),
                             ("generic kernel, folded forward", _hip.MLP_SAMPLE_GENERIC_KERNEL),
                             ("generic kernel, layer by layer", _hip.MLP_SAMPLE_GENERIC_KERNEL | _hip.MLP_SAMPLE_UNFOLDED)):
        gen = P["Langevin"](P["Noise"](**noise_kw), P["Sampling"](**sampling_kw, rng_mode="device", seed=3131, fused_score_network=True),
                            net.to(cuda))
        gen.fused_sampler_options = options
        summary(f"periodic-well MLP (out.X = -0.3 sin 2 pi (x - site)), configs[1]'s job, persistent fused kernel: {variant}", g,
                [gen.sample(int(g["batch"]), cuda).X.cpu().numpy() for _ in range(len(g["seeds"]))])
    for probe in ("zero_score", "score_x0.9", "score_x0.97", "no_corrector", "sigma_min_1e-3"):
        keys = T.scalars_of(g)
        print(f"   reference-side wrong sampler {probe:14s}: first call " + ", ".join(f"{k} {float(g[f'probe/{probe}/{k}']):.4f}" for k in keys) +
              f" | {int(g['probe_calls'])} calls pooled " + ", ".join(f"{k} {float(g[f'probe_pooled/{probe}/{k}']):.4f}" for k in keys) +
              f"  -> {'rejected' if T.probe_fails(g, probe) else 'not seen'}")
