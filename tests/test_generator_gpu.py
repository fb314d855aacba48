"""GPU parity tests of the generators (the HIP path behind the reference's generator API).

  * reference-RNG mode, draws replayed from the golden fixtures -> final and per-step compositions against the
    reference's own recorded trajectories (A exact; X <= 1e-5 rel-L2 on the torus, the tolerance of north_star);
  * the same runs against the CPU oracle: with the echo network (whose forward is exact on both sides) every
    float must be BIT-IDENTICAL; with real networks the GPU/CPU forward differs by ~1e-6 and the tolerance applies;
  * device-RNG (Philox) mode against the oracle evaluating the same specification; graph replay == eager;
  * BASELINE-size runs checked through size-independent properties.
"""
import numpy as np
import pytest
import torch

import cases
import nets
from conftest import load_golden, torus_rel_l2
from oracle import reference_sampler as RS

# MDX_FUZZ=k multiplies the number of seeds of the random-configuration tests below (a one-off wider sweep: MDX_FUZZ=10
# python -m pytest tests/test_generator_gpu.py -m gpu -k random); the suite runs with 1.
import os
FUZZ = max(1, int(os.environ.get("MDX_FUZZ", "1")))

pytestmark = pytest.mark.gpu


def _pkg():
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.constrained_langevin_generator import \
        ConstrainedLangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.noise_sources import ReferenceOrderNoise
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.sampling_constraint import SamplingConstraint
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    return dict(Langevin=LangevinGenerator, Constrained=ConstrainedLangevinGenerator, RefNoise=ReferenceOrderNoise,
                Sampling=PredictorCorrectorSamplingParameters, Constraint=SamplingConstraint, Noise=NoiseParameters)


def _replayed(fixture):
    P = _pkg()

    class Replayed(P["RefNoise"]):
        def __init__(self, g):
            self.inner = RS.ReplayNoise(g)

        def rand(self, *shape):
            shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else shape
            return torch.from_numpy(self.inner.rand(*shape))

        def randn(self, *shape):
            shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else shape
            return torch.from_numpy(self.inner.randn(*shape))

    return Replayed(fixture)


def _build(name, table, cuda, fixture=None, constraint=None, **extra):
    import warnings
    P = _pkg()
    noise_kw, sampling_kw, netf = table[name]
    skw = dict(sampling_kw)
    skw.update(extra)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar, spar = P["Noise"](**noise_kw), P["Sampling"](**skw)
    if netf is None:
        net_gpu, net_cpu = nets.fake_net(spar.num_atom_types), nets.fake_net(spar.num_atom_types)
    else:
        torch.manual_seed(1234)          # same random-init weights for every build of a case
        net_gpu = netf(None)
        net_cpu = netf(nets.oracle_edge_builder)
        if fixture is not None:
            nets.load_fixture_weights(net_gpu, fixture)
        net_cpu.load_state_dict(net_gpu.state_dict())
    net_gpu = net_gpu.to(cuda)
    if constraint is not None:
        gen = P["Constrained"](npar, spar, net_gpu, constraint)
    else:
        gen = P["Langevin"](npar, spar, net_gpu)
    return gen, npar, spar, net_cpu


def _np(axl):
    return RS.AXL(A=axl.A.cpu().numpy(), X=axl.X.cpu().numpy(), L=axl.L.cpu().numpy())


# Free-running trajectories are compared at north_star's 1e-5 wherever the sampler's map is not expanding.  The MLP
# template configuration (exponential schedule, sigma_min 1e-4, default corrector_step_epsilon 2e-5) is: its first
# correctors multiply the score by eps_i/sigma_i ~ 50, and the REFERENCE ITSELF turns a 1e-7 perturbation of the
# initial coordinates into 7.7e-4 after 20 steps (tests/test_oracle_golden.py::test_conditioning_of_reference_map).
# There the GPU (whose network forward differs from the CPU forward in the last bit, ~3e-8) is held to that
# conditioning in free run, and to 1e-5 per step in test_teacher_forced_steps below.
FREE_RUN_TOLERANCE = {"traj_mlp_c1": 5e-3}


@pytest.mark.parametrize("name", list(cases.TRAJECTORIES))
def test_teacher_forced_steps(cuda, name):
    """Every predictor / corrector step, started from the composition the REFERENCE recorded at that step and fed
    the reference's draws: output within 1e-5 rel-L2 (torus) of the reference's recorded output, atom types exact."""
    g = load_golden(name + ".npz")
    gen, npar, spar, _ = _build(name, cases.TRAJECTORIES, cuda, fixture=g)
    gen.noise_source = _replayed(g)
    B, M = int(g["batch"]), spar.number_of_corrector_steps

    def axl(prefix, k):
        return RS.AXL(A=torch.from_numpy(g[prefix + "_A"][k]).to(cuda), X=torch.from_numpy(g[prefix + "_X"][k]).to(cuda),
                      L=torch.from_numpy(g[prefix + "_L"][k]).to(cuda))

    with torch.no_grad():
        gen._prepare(cuda)
        gen._begin_call(cuda)
        gen.initialize(B, cuda)                     # consumes the initial draws
        forces = torch.zeros(B, spar.number_of_atoms, 3, device=cuda)
        worst = 0.0
        for k, index in enumerate(g["pred_index"]):
            out = gen.predictor_step(axl("pred_composition_i", k), int(index), forces)
            assert np.array_equal(out.A.cpu().numpy(), g["pred_composition_im1_A"][k]), (name, "pred", k)
            worst = max(worst, torus_rel_l2(out.X.cpu().numpy(), g["pred_composition_im1_X"][k]))
            np.testing.assert_allclose(out.L.cpu().numpy(), g["pred_composition_im1_L"][k], rtol=1e-5, atol=1e-6)
            for m in range(M):
                kk = k * M + m
                out = gen.corrector_step(axl("corr_composition_i", kk), int(index) - 1, forces, m)
                assert np.array_equal(out.A.cpu().numpy(), g["corr_corrected_composition_i_A"][kk]), (name, "corr", kk)
                worst = max(worst, torus_rel_l2(out.X.cpu().numpy(), g["corr_corrected_composition_i_X"][kk]))
    assert gen.noise_source.inner.exhausted()
    assert worst < 1e-5, f"{name}: worst per-step rel-L2 {worst:.2e}"


@pytest.mark.parametrize("name", list(cases.TRAJECTORIES))
def test_reference_mode_against_golden_and_oracle(cuda, name):
    tol = FREE_RUN_TOLERANCE.get(name, 1e-5)
    g = load_golden(name + ".npz")
    gen, npar, spar, net_cpu = _build(name, cases.TRAJECTORIES, cuda, fixture=g, record_samples=True,
                                      record_samples_corrector_steps=True)
    gen.noise_source = _replayed(g)
    with torch.no_grad():
        out = _np(gen.sample(int(g["batch"]), cuda))
    assert gen.noise_source.inner.exhausted()
    # against the reference
    assert np.array_equal(out.A, g["final_A"])
    assert torus_rel_l2(out.X, g["final_X"]) < tol
    np.testing.assert_allclose(out.L, g["final_L"], rtol=1e-5, atol=1e-6)
    rec = gen.sample_trajectory_recorder._internal_data
    assert [e["time_step_index"] for e in rec["predictor_step"]] == list(g["pred_index"])
    for k, e in enumerate(rec["predictor_step"]):
        assert np.array_equal(e["composition_im1"].A.numpy(), g["pred_composition_im1_A"][k])
        assert torus_rel_l2(e["composition_im1"].X.numpy(), g["pred_composition_im1_X"][k]) < tol
    if "corr_index" in g.files:
        assert [e["time_step_index"] for e in rec["corrector_step"]] == list(g["corr_index"])
    # against the oracle on the same draws
    ora = RS.OracleLangevinGenerator(npar, spar, net_cpu, noise=RS.ReplayNoise(g)).sample(int(g["batch"]))
    assert np.array_equal(out.A, ora.A)
    if cases.TRAJECTORIES[name][2] is None:      # echo network: exact forward on both sides => bit-identical floats
        assert np.array_equal(out.X.view(np.int32), ora.X.view(np.int32))
        assert np.array_equal(out.L.view(np.int32), ora.L.view(np.int32))
    else:
        assert torus_rel_l2(out.X, ora.X) < tol


@pytest.mark.parametrize("name", list(cases.REPAINT))
def test_repaint_reference_mode(cuda, name):
    P = _pkg()
    g = load_golden(name + ".npz")
    nat = cases.REPAINT[name][1]["num_atom_types"]
    constraint = P["Constraint"](elements=["Si", "Ge"][:nat],
                                 constrained_relative_coordinates=torch.from_numpy(g["constrained_relative_coordinates"]),
                                 constrained_atom_types=torch.from_numpy(g["constrained_atom_types"]),
                                 constrained_indices=torch.from_numpy(g["constrained_indices"]))
    gen, npar, spar, net_cpu = _build(name, cases.REPAINT, cuda, fixture=g, constraint=constraint)
    gen.noise_source = _replayed(g)
    with torch.no_grad():
        out = _np(gen.sample(int(g["batch"]), cuda))
    assert gen.noise_source.inner.exhausted()
    assert np.array_equal(out.A, g["final_A"])
    assert torus_rel_l2(out.X, g["final_X"]) < 1e-5
    idx = g["constrained_indices"]
    assert np.array_equal(out.X[:, idx], np.broadcast_to(g["constrained_relative_coordinates"], out.X[:, idx].shape))
    assert np.array_equal(out.A[:, idx], np.broadcast_to(g["constrained_atom_types"], out.A[:, idx].shape))


DEVICE_CASES = ["traj_fake_c2", "traj_fake_c3_m2", "traj_fake_c5_nogreedy", "traj_fake_c5_test",
                "traj_fake_free_lattice", "traj_mlp_c1", "traj_mlp_c3", "traj_egnn_fc", "traj_egnn_rc"]


@pytest.mark.parametrize("name", DEVICE_CASES)
def test_device_rng_mode_against_oracle(cuda, name):
    seed, batch = 20250815, 6
    gen, npar, spar, net_cpu = _build(name, cases.TRAJECTORIES, cuda, rng_mode="device", seed=seed)
    with torch.no_grad():
        first = _np(gen.sample(batch, cuda))
        second = _np(gen.sample(batch, cuda))          # second call -> Philox call index 1
    for call, out in enumerate((first, second)):
        ora = RS.OracleLangevinGenerator(npar, spar, net_cpu, noise=RS.PhiloxNoise(seed, call)).sample(batch)
        assert np.array_equal(out.A, ora.A), (name, call)
        if cases.TRAJECTORIES[name][2] is None:
            assert np.array_equal(out.X.view(np.int32), ora.X.view(np.int32))
            assert np.array_equal(out.L.view(np.int32), ora.L.view(np.int32))
        else:
            assert torus_rel_l2(out.X, ora.X) < FREE_RUN_TOLERANCE.get(name, 1e-5)
        assert (out.A != spar.num_atom_types).all()
    assert not np.array_equal(first.X, second.X)


@pytest.mark.parametrize("name", ["traj_repaint_fake", "traj_repaint_mlp"])
def test_repaint_device_rng_against_oracle(cuda, name):
    P = _pkg()
    rng = np.random.default_rng(3)
    nat = cases.REPAINT[name][1]["num_atom_types"]
    cx = rng.random((4, 3), dtype=np.float32)
    ca = rng.integers(0, nat, 4)
    cidx = np.array([5, 0, 2, 7])
    constraint = P["Constraint"](elements=["Si", "Ge"][:nat], constrained_relative_coordinates=torch.from_numpy(cx),
                                 constrained_atom_types=torch.from_numpy(ca), constrained_indices=torch.from_numpy(cidx))
    gen, npar, spar, net_cpu = _build(name, cases.REPAINT, cuda, constraint=constraint, rng_mode="device", seed=11)
    with torch.no_grad():
        out = _np(gen.sample(5, cuda))
    ora = RS.OracleLangevinGenerator(npar, spar, net_cpu, noise=RS.PhiloxNoise(11, 0),
                                     constraint=dict(constrained_relative_coordinates=cx, constrained_atom_types=ca,
                                                     constrained_indices=cidx)).sample(5)
    assert np.array_equal(out.A, ora.A)
    if cases.REPAINT[name][2] is None:
        assert np.array_equal(out.X.view(np.int32), ora.X.view(np.int32))
    else:
        assert torus_rel_l2(out.X, ora.X) < 1e-5
    assert np.array_equal(out.X[:, cidx], np.broadcast_to(cx, (5, 4, 3)))


@pytest.mark.parametrize("name", ["traj_fake_c3_m2", "traj_mlp_c1", "traj_mlp_c3"])
def test_graph_replay_equals_eager(cuda, name):
    outs = []
    for use_graph in (False, True):
        gen, *_ = _build(name, cases.TRAJECTORIES, cuda, rng_mode="device", seed=5, use_hip_graph=use_graph)
        torch.manual_seed(1)
        with torch.no_grad():
            outs.append(_np(gen.sample(32, cuda)))
    assert np.array_equal(outs[0].A, outs[1].A)
    assert np.array_equal(outs[0].X.view(np.int32), outs[1].X.view(np.int32))   # same kernels, same draws


@pytest.mark.parametrize("name", ["traj_mlp_c3", "traj_egnn_rc", "traj_repaint_mlp"])
def test_captured_iteration_is_reused_across_sample_calls(cuda, name):
    """use_hip_graph: the iteration is captured by the FIRST sample() call of a shape and replayed by the later ones (the Philox
    call index lives in a device word that _begin_call rewrites; the start composition is copied into the graph's buffers).
    Three consecutive calls equal the eager generator's three calls bit for bit, differ from one another, do not share storage,
    and run on ONE IterationLoop; a changed network parameter or batch size captures anew."""
    P = _pkg()
    table = cases.REPAINT if "repaint" in name else cases.TRAJECTORIES
    constraint = None
    if "repaint" in name:
        constraint = P["Constraint"](elements=["Si"], constrained_relative_coordinates=torch.rand(3, 3),
                                     constrained_atom_types=torch.zeros(3, dtype=torch.long))
    extra = dict(repaint_resampling_steps=1) if "repaint" in name else {}
    outs = {}
    for use_graph in (False, True):
        gen, *_ = _build(name, table, cuda, constraint=constraint, rng_mode="device", seed=41, use_hip_graph=use_graph, **extra)
        with torch.no_grad():
            calls = [gen.sample(7, cuda) for _ in range(3)]
        if use_graph:
            loop = gen._buffers["graph_loop"]
            kept = [c.X.clone() for c in calls]
            with torch.no_grad():
                again = gen.sample(7, cuda)
                assert gen._buffers["graph_loop"] is loop                        # a fourth call: still the first capture
                assert all(torch.equal(c.X, k) for c, k in zip(calls, kept))    # earlier results are not the graph's buffers
                assert len({c.X.data_ptr() for c in calls + [again]}) == 4
                other = gen.sample(5, cuda)                                      # another batch size: a capture of its own
                assert gen._buffers["graph_loop"] is not loop and other.X.shape[0] == 5
                loop = gen._buffers["graph_loop"]
                p0 = next(gen.axl_network.parameters())
                p0.mul_(1.0)                                                     # in-place: bumps the parameter's version
                gen.sample(5, cuda)
                assert gen._buffers["graph_loop"] is not loop
        outs[use_graph] = [_np(c) for c in calls]
    for a, b in zip(outs[False], outs[True]):
        assert np.array_equal(a.A, b.A) and np.array_equal(a.X.view(np.int32), b.X.view(np.int32))
    assert not np.array_equal(outs[True][0].X, outs[True][1].X) and not np.array_equal(outs[True][1].X, outs[True][2].X)


def test_captured_iteration_is_reused_with_an_unindexed_device(cuda):
    """`--device cuda` (the CLI's default) hands the generator torch.device("cuda"), which does not compare equal to cuda:0: the
    device is normalised once, so the tables, the status / call words and the captured iteration are kept across calls, and the
    results equal those of the same calls with cuda:0."""
    plain = torch.device("cuda")
    outs = {}
    for device in (plain, cuda):
        gen, *_ = _build("traj_mlp_c3", cases.TRAJECTORIES, cuda, rng_mode="device", seed=43, use_hip_graph=True)
        with torch.no_grad():
            first = gen.sample(6, device)
            loop, word, tables = gen._buffers["graph_loop"], gen._call_word, gen._scheduler
            second = gen.sample(6, device)
        assert gen._buffers["graph_loop"] is loop and gen._call_word is word and gen._scheduler is tables
        outs[device] = (_np(first), _np(second))
    for a, b in zip(outs[plain], outs[cuda]):
        assert np.array_equal(a.A, b.A) and np.array_equal(a.X.view(np.int32), b.X.view(np.int32))
    assert not np.array_equal(outs[plain][0].X, outs[plain][1].X)


def test_graph_replay_repaint(cuda):
    P = _pkg()
    constraint = P["Constraint"](elements=["Si"], constrained_relative_coordinates=torch.rand(3, 3),
                                 constrained_atom_types=torch.zeros(3, dtype=torch.long))
    outs = []
    for use_graph in (False, True):
        gen, *_ = _build("traj_repaint_mlp", cases.REPAINT, cuda, constraint=constraint, rng_mode="device", seed=9,
                         use_hip_graph=use_graph)
        with torch.no_grad():
            outs.append(_np(gen.sample(16, cuda)))
    assert np.array_equal(outs[0].A, outs[1].A)
    assert np.array_equal(outs[0].X.view(np.int32), outs[1].X.view(np.int32))


def _resampling_constraint(P, nat):
    rng = np.random.default_rng(3)
    cx = rng.random((4, 3), dtype=np.float32)
    ca = rng.integers(0, nat, 4)
    cidx = np.array([5, 0, 2, 7])
    constraint = P["Constraint"](elements=["Si", "Ge"][:nat], constrained_relative_coordinates=torch.from_numpy(cx),
                                 constrained_atom_types=torch.from_numpy(ca), constrained_indices=torch.from_numpy(cidx))
    return constraint, dict(constrained_relative_coordinates=cx, constrained_atom_types=ca, constrained_indices=cidx)


@pytest.mark.parametrize("rng_mode", ["device", "reference"])
@pytest.mark.parametrize("name", ["traj_repaint_fake", "traj_repaint_mlp"])
def test_repaint_resampling_against_oracle(cuda, name, rng_mode):
    """Build-only `repaint_resampling_steps` (BASELINE configs[4] "with resampling"; the reference has no such loop):
    GPU generator vs the oracle's restatement of the same specification, in both RNG modes."""
    P = _pkg()
    nat = cases.REPAINT[name][1]["num_atom_types"]
    constraint, as_dict = _resampling_constraint(P, nat)
    outs = {}
    for steps in (0, 2):
        gen, npar, spar, net_cpu = _build(name, cases.REPAINT, cuda, constraint=constraint, rng_mode=rng_mode, seed=11,
                                          repaint_resampling_steps=steps)
        torch.manual_seed(77)
        with torch.no_grad():
            out = _np(gen.sample(5, cuda))
        torch.manual_seed(77)
        noise = RS.PhiloxNoise(11, 0) if rng_mode == "device" else RS.TorchCpuNoise()
        ora = RS.OracleLangevinGenerator(npar, spar, net_cpu, noise=noise, constraint=as_dict).sample(5)
        assert np.array_equal(out.A, ora.A), steps
        if cases.REPAINT[name][2] is None:
            assert np.array_equal(out.X.view(np.int32), ora.X.view(np.int32))
        else:
            assert torus_rel_l2(out.X, ora.X) < 1e-5
        cidx = as_dict["constrained_indices"]
        assert np.array_equal(out.X[:, cidx], np.broadcast_to(as_dict["constrained_relative_coordinates"], (5, 4, 3)))
        assert (out.A != nat).all()
        outs[steps] = out
    assert not np.array_equal(outs[0].X, outs[2].X)        # the resampling passes really ran


def test_graph_replay_repaint_resampling(cuda):
    P = _pkg()
    constraint, _ = _resampling_constraint(P, 1)
    outs = []
    for use_graph in (False, True):
        gen, *_ = _build("traj_repaint_mlp", cases.REPAINT, cuda, constraint=constraint, rng_mode="device", seed=9,
                         use_hip_graph=use_graph, repaint_resampling_steps=1)
        with torch.no_grad():
            outs.append(_np(gen.sample(16, cuda)))
    assert np.array_equal(outs[0].A, outs[1].A)
    assert np.array_equal(outs[0].X.view(np.int32), outs[1].X.view(np.int32))


def test_batch_driver_against_golden(cuda):
    from diffusion_for_multi_scale_molecular_dynamics_amd.sampling.diffusion_sampling import create_batch_of_samples
    g = load_golden("batch_of_samples.npz")
    P = _pkg()
    npar = P["Noise"](**cases.noise_ns(6))
    spar = P["Sampling"](**dict(cases.sampling_ns(8, 1), number_of_samples=7, sample_batchsize=3))
    gen = P["Langevin"](npar, spar, nets.fake_net(1).to(cuda))
    gen.noise_source = _replayed(g)
    with torch.no_grad():
        batch = create_batch_of_samples(gen, spar, cuda)
    assert np.array_equal(batch["original_axl"].A.cpu().numpy(), g["A"])
    assert torus_rel_l2(batch["original_axl"].X.cpu().numpy(), g["X"]) < 1e-5
    np.testing.assert_allclose(batch["cartesian_positions"].cpu().numpy(), g["cartesian_positions"], rtol=1e-5, atol=1e-5)


def test_last_step_mask_is_reported(cuda):
    """small_epsilon is also the floor of the class probabilities (reference quirk 7): with a floor of 0.5 the MASK
    class keeps a large posterior at the last step, so some atoms stay masked.  The reference asserts inside the
    last predictor step (langevin_generator.py:616-620); the build raises after its single end-of-run status read."""
    P = _pkg()
    npar = P["Noise"](**cases.noise_ns(3))
    spar = P["Sampling"](**cases.sampling_ns(8, 1, greedy=False, one=False, eps=0.5), rng_mode="device", seed=1)
    gen = P["Langevin"](npar, spar, nets.fake_net(1).to(cuda))
    with pytest.raises(AssertionError, match="MASKED atoms"):
        gen.sample(64, cuda)
    # the status word is cleared by the read: a healthy run on the same generator passes afterwards
    gen.small_epsilon = 1e-8
    assert (gen.sample(4, cuda).A == 0).all()


@pytest.mark.parametrize("config", ["C2", "C4"])
def test_baseline_size_properties(cuda, config):
    """BASELINE-size batches (few steps): invariants that do not need the oracle at full size."""
    P = _pkg()
    if config == "C2":      # Si 1x1x1, MLP, B=1024
        B, N, nat, M = 1024, 8, 1, 1
        net = nets.mlp_net(N, nat, seed=1234)
        noise_kw = cases.noise_ns(12, sigma_min=1e-4, sigma_max=0.25)
        skw = cases.sampling_ns(N, nat, M=M)
    else:                   # SiGe 2x2x2 shard, EGNN radial cutoff, two atom types, B=512 (reduced width/steps)
        B, N, nat, M = 512, 64, 2, 2
        net = nets.egnn_net(nat, "radial_cutoff", 7.5, hidden=32, seed=1234)
        noise_kw = cases.noise_ns(4, **cases.LIN)
        skw = cases.sampling_ns(N, nat, M=M, cell=[11.084] * 3)
    gen = P["Langevin"](P["Noise"](**noise_kw), P["Sampling"](**skw, rng_mode="device", seed=77), net.to(cuda))
    with torch.no_grad():
        a = gen.sample(B, cuda)
        b = gen.sample(B, cuda)
    for out in (a, b):
        assert out.X.shape == (B, N, 3) and out.A.shape == (B, N)
        assert bool(((out.X >= 0) & (out.X < 1)).all())
        assert bool(((out.A >= 0) & (out.A < nat)).all())            # fully unmasked
        assert bool(torch.isfinite(out.X).all())
    assert not torch.equal(a.X, b.X)
    # same seed, fresh generator => same stream (counter-based RNG: a pure function of (seed, call, step, atom))
    gen2 = P["Langevin"](P["Noise"](**noise_kw), P["Sampling"](**skw, rng_mode="device", seed=77), net.to(cuda))
    with torch.no_grad():
        a2 = gen2.sample(B, cuda)
    assert torch.equal(a.A, a2.A)
    if config == "C2":
        assert torch.equal(a.X, a2.X)


def test_cli_end_to_end(cuda, tmp_path):
    """Counterpart of the reference's tests/test_sample_diffusion.py:197-237 on the GPU path."""
    import yaml
    from diffusion_for_multi_scale_molecular_dynamics_amd import sample_diffusion
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.sampling_constraint import (
        SamplingConstraint, write_sampling_constraint)
    cfg = dict(noise=dict(total_time_steps=10, sigma_min=1e-4, sigma_max=0.25),
               sampling=dict(algorithm="predictor_corrector", spatial_dimension=3, number_of_atoms=8,
                             number_of_samples=12, sample_batchsize=5, num_atom_types=1,
                             number_of_corrector_steps=1, record_samples=True, use_fixed_lattice_parameters=True,
                             cell_dimensions=[5.43, 5.43, 5.43]),
               elements=["Si"],
               model=dict(score_network=dict(architecture="mlp", number_of_atoms=8, num_atom_types=1,
                                             n_hidden_dimensions=2, hidden_dimensions_size=16,
                                             relative_coordinates_embedding_dimensions_size=8,
                                             noise_embedding_dimensions_size=4, time_embedding_dimensions_size=4,
                                             atom_type_embedding_dimensions_size=1,
                                             lattice_parameters_embedding_dimensions_size=1)))
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(cfg))
    constraint = SamplingConstraint(elements=["Si"], constrained_relative_coordinates=torch.rand(3, 3),
                                    constrained_atom_types=torch.zeros(3, dtype=torch.long))
    write_sampling_constraint(constraint, tmp_path / "constraint.pkl")
    for sub, extra in (("plain", []), ("repaint", ["--path_to_sampling_constraint_data_pickle",
                                                   str(tmp_path / "constraint.pkl")])):
        out = tmp_path / sub
        sample_diffusion.main(["--config", str(tmp_path / "config.yaml"), "--output", str(out), "--device", "cuda",
                               "--random_init_seed", "3"] + extra)
        samples = torch.load(out / "samples.pt", weights_only=False)
        assert samples["cartesian_positions"].shape == (12, 8, 3)
        assert samples["original_axl"].A.shape == (12, 8) and samples["original_axl"].L.shape == (12, 6)
        assert (out / "trajectories.pt").exists() and (out / "config_backup.yaml").exists()
        if sub == "repaint":
            x = samples["original_axl"].X[:, :3].cpu()
            assert torch.equal(x, constraint.constrained_relative_coordinates.expand(12, 3, 3))
    # --reference_pickles: the same run, its files naming the REFERENCE's AXL class (read back through the tolerant loader)
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils import reference_pickles
    sample_diffusion.main(["--config", str(tmp_path / "config.yaml"), "--output", str(tmp_path / "for_reference"), "--device", "cuda",
                           "--random_init_seed", "3", "--reference_pickles"])
    for name in ("samples.pt", "trajectories.pt"):
        raw = (tmp_path / "for_reference" / name).read_bytes()
        assert b"diffusion_for_multi_scale_molecular_dynamics_amd" not in raw and b"diffusion_for_multi_scale_molecular_dynamics" in raw
    for_reference = reference_pickles.load(tmp_path / "for_reference" / "samples.pt")
    assert torch.equal(for_reference["original_axl"].X.cpu(),
                       torch.load(tmp_path / "plain" / "samples.pt", weights_only=False)["original_axl"].X.cpu())
    # the `force_field:` block (src/sample_diffusion.py:132-139): a positive cutoff wraps the network; a zero cutoff never reaches
    # the reference's "using original network" branch -- ForceFieldParameters refuses it (force_field_augmented_score_network.py:
    # 34-38) -- and does not here; an `oracle:` block is reported as out of scope, the samples are still written
    plain = torch.load(tmp_path / "plain" / "samples.pt", weights_only=False)["original_axl"].X
    (tmp_path / "ff.yaml").write_text(yaml.safe_dump(dict(cfg, force_field=dict(radial_cutoff=2.5, strength=5.0), oracle=dict(name="lammps"))))
    sample_diffusion.main(["--config", str(tmp_path / "ff.yaml"), "--output", str(tmp_path / "ff"), "--device", "cuda",
                           "--random_init_seed", "3"])
    x = torch.load(tmp_path / "ff" / "samples.pt", weights_only=False)["original_axl"].X
    assert x.shape == plain.shape and not torch.equal(x, plain)
    log = (tmp_path / "ff" / "console.log").read_text()
    assert "excluding Force Field" in log and "energies.pt is not" in log and not (tmp_path / "ff" / "energies.pt").exists()
    (tmp_path / "ff_zero.yaml").write_text(yaml.safe_dump(dict(cfg, force_field=dict(radial_cutoff=0.0, strength=5.0))))
    with pytest.raises(AssertionError, match="greater than zero"):
        sample_diffusion.main(["--config", str(tmp_path / "ff_zero.yaml"), "--output", str(tmp_path / "ff_zero"), "--device", "cuda",
                               "--random_init_seed", "3"])
    # the `elements` list is validated (data/element_types.py:35-38)
    for elements, message in ((["Si", "Si"], "should be unique"), (["NULL_ELEMENT_FOR_PADDING"], "is reserved")):
        (tmp_path / "bad.yaml").write_text(yaml.safe_dump(dict(cfg, elements=elements)))
        with pytest.raises(AssertionError, match=message):
            sample_diffusion.main(["--config", str(tmp_path / "bad.yaml"), "--output", str(tmp_path / "bad"), "--device", "cuda",
                                   "--random_init_seed", "3"])


# -------------------------------------------------------------------------------------------------------------
# fused MLP score network: forward against the PyTorch module, persistent sampler against the per-step path
# -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_atoms,nat,hidden,n_hidden", [(8, 1, 64, 3), (8, 2, 64, 3), (5, 4, 48, 2), (20, 1, 96, 4),
                                                         (64, 2, 128, 3)])
def test_fused_mlp_forward_against_torch(cuda, n_atoms, nat, hidden, n_hidden):
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    torch.manual_seed(n_atoms + nat)
    net = nets.mlp_net(n_atoms, nat, hidden=hidden, n_hidden=n_hidden).to(cuda)
    B = 37
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, nat + 1, (B, n_atoms), device=cuda),
                                        X=torch.rand(B, n_atoms, 3, device=cuda),
                                        L=torch.tensor([5.43, 5.5, 5.6, 0, 0, 0.0], device=cuda).repeat(B, 1)),
             TIME: torch.rand(B, 1, device=cuda), NOISE: torch.rand(B, 1, device=cuda) * 0.25,
             CARTESIAN_FORCES: torch.zeros(B, n_atoms, 3, device=cuda)}
    with torch.no_grad():
        want = net(batch, conditional=False)
    pack = kernels.MlpPack(net, cuda)
    comp = batch[NOISY_AXL_COMPOSITION]
    logits, sx, sl = kernels.mlp_forward(pack, comp.A, comp.X, comp.L, batch[TIME], batch[NOISE])
    assert torch.isinf(logits[..., -1]).all()
    # float32 forward with a different summation order and an exactly periodic cos/sin: 1e-5 relative to the output scale
    for got, ref in ((sx, want.X), (sl, want.L), (logits[..., :-1], want.A[..., :-1])):
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-7


@pytest.mark.parametrize("name", ["traj_mlp_c3", "traj_mlp_c1"])
def test_fused_sampler_against_per_step_path_and_oracle(cuda, name):
    """One persistent launch for the whole trajectory == the per-step kernels with the PyTorch forward (same Philox
    draws; the forwards differ by float32 rounding only) == the CPU oracle."""
    seed, batch = 99, 40
    outs = {}
    for fused in (False, True):
        gen, npar, spar, net_cpu = _build(name, cases.TRAJECTORIES, cuda, rng_mode="device", seed=seed,
                                          fused_score_network=fused)
        with torch.no_grad():
            outs[fused] = _np(gen.sample(batch, cuda))
    ora = RS.OracleLangevinGenerator(npar, spar, net_cpu, noise=RS.PhiloxNoise(seed, 0)).sample(batch)
    tol = FREE_RUN_TOLERANCE.get(name, 1e-5)
    assert np.array_equal(outs[True].A, outs[False].A) and np.array_equal(outs[True].A, ora.A)
    assert torus_rel_l2(outs[True].X, outs[False].X) < tol
    assert torus_rel_l2(outs[True].X, ora.X) < tol


def test_fused_sampler_segments_compose(cuda):
    """Running the loop as 3 launches over consecutive index ranges == one launch (state round-trips through HBM)."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    gen, npar, spar, _ = _build("traj_mlp_c3", cases.TRAJECTORIES, cuda, rng_mode="device", seed=5,
                                fused_score_network=True)
    with torch.no_grad():
        gen._prepare(cuda)
        gen._begin_call(cuda)
        start = gen.initialize(16, cuda)
        whole = gen._sample_fused(start, 16, 0)
        comp = RS.AXL(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
        sched, pack = gen._prepare(cuda), gen.fused_pack(cuda)
        for first, n in ((16, 5), (11, 10), (1, 1)):
            kernels.mlp_pc_sample(sched, pack, gen._flags(True), 2, False, first, n, gen._rng(0), comp.A, comp.X, comp.L,
                                  gen._status, workspace=gen._noise_workspace)
    assert torch.equal(whole.A, comp.A) and torch.equal(whole.X, comp.X)


@pytest.mark.parametrize("name,in_corrector", [("traj_mlp_c3", False), ("traj_mlp_c1", False), ("traj_mlp_c3", True)])
def test_fused_sampler_predrawn_noise_equals_in_kernel_draws(cuda, name, in_corrector, monkeypatch):
    """The noise pre-pass (chip-filling kernel + workspace) and the in-kernel draws evaluate the same Philox
    specification: bit-identical trajectories; a workspace that holds only 3 iterations splits the segment."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    gen, npar, spar, _ = _build(name, cases.TRAJECTORIES, cuda, rng_mode="device", seed=5, fused_score_network=True,
                                atom_type_transition_in_corrector=in_corrector)
    T, M = npar.total_time_steps, spar.number_of_corrector_steps
    with torch.no_grad():
        gen._prepare(cuda)
        gen._begin_call(cuda)
        start = gen.initialize(19, cuda)
        sched, pack = gen._prepare(cuda), gen.fused_pack(cuda)
        outs = []
        for mode in ("in_kernel", "predrawn", "predrawn_small_workspace"):
            comp = RS.AXL(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
            workspace = None if mode == "in_kernel" else kernels.NoiseWorkspace()
            if mode == "predrawn_small_workspace":
                per_iteration = kernels.lib().mdx_mlp_pc_sample_workspace_floats(pack.c_struct, M, int(in_corrector), 1, 19)
                monkeypatch.setattr(kernels, "NOISE_WORKSPACE_MAX_FLOATS", 3 * per_iteration)
                monkeypatch.setattr(kernels, "NOISE_WORKSPACE_MIN_ITERATIONS", 1)
            kernels.mlp_pc_sample(sched, pack, gen._flags(True), M, in_corrector, T, T, gen._rng(0), comp.A, comp.X,
                                  comp.L, gen._status, workspace=workspace)
            if mode == "predrawn_small_workspace":
                assert workspace.buffer.numel() == 3 * per_iteration          # the segment really was split
            outs.append(comp)
        monkeypatch.undo()
        rec0 = spar.number_of_atoms * (3 + gen.num_classes + 1) + 8      # z | gumbel | u | tabulated posterior (8)
        rec1 = rec0 if in_corrector else spar.number_of_atoms * 3
        floats = T * 19 * (rec0 + M * rec1)
        assert kernels.lib().mdx_mlp_pc_sample_workspace_floats(pack.c_struct, M, int(in_corrector), T, 19) == floats
        assert kernels.NoiseWorkspace().get(pack, M, in_corrector, T, 19, cuda).numel() >= floats
    for other in outs[1:]:
        assert torch.equal(outs[0].A, other.A)
        assert torch.equal(outs[0].X.view(torch.int32), other.X.view(torch.int32))
    assert (outs[0].A != spar.num_atom_types).all()


def test_fused_needs_mlp_and_device_rng(cuda):
    from diffusion_for_multi_scale_molecular_dynamics_amd._hip import MdxError
    gen, *_ = _build("traj_fake_c2", cases.TRAJECTORIES, cuda, rng_mode="device", seed=1, fused_score_network=True)
    with pytest.raises(MdxError, match="MLPScoreNetwork"):
        gen.sample(2, cuda)
    gen, *_ = _build("traj_mlp_c3", cases.TRAJECTORIES, cuda, fused_score_network=True)     # reference RNG
    with pytest.raises(MdxError, match="rng_mode='device'"):
        gen.sample(2, cuda)


# -------------------------------------------------------------------------------------------------------------
# SURVEY 8(f) "next" rows: adaptive corrector, force-field augmentation, atom-type update recording
# -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(cases.ADAPTIVE))
def test_adaptive_corrector_against_golden_and_oracle(cuda, name):
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.adaptive_corrector import AdaptiveCorrectorGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.instantiate_generator import instantiate_generator
    P = _pkg()
    g = load_golden(name + ".npz")
    noise_kw, sampling_kw, netf = cases.ADAPTIVE[name]
    npar, spar = P["Noise"](**noise_kw), P["Sampling"](**sampling_kw)
    torch.manual_seed(1234)
    net = nets.fake_net(spar.num_atom_types) if netf is None else nets.load_fixture_weights(netf(None), g)
    import copy
    net_cpu = copy.deepcopy(net)
    gen = instantiate_generator(spar, npar, net.to(cuda), None)
    assert type(gen) is AdaptiveCorrectorGenerator
    gen.noise_source = _replayed(g)
    with torch.no_grad():
        out = _np(gen.sample(int(g["batch"]), cuda))
    assert gen.noise_source.inner.exhausted()
    assert np.array_equal(out.A, g["final_A"])
    assert torus_rel_l2(out.X, g["final_X"]) < 1e-5
    # device RNG against the oracle
    spar2 = P["Sampling"](**dict(sampling_kw, rng_mode="device", seed=17))
    gen2 = AdaptiveCorrectorGenerator(npar, spar2, net)
    with torch.no_grad():
        dev_out = _np(gen2.sample(6, cuda))
    ora = RS.OracleAdaptiveCorrectorGenerator(npar, spar2, net_cpu, noise=RS.PhiloxNoise(17, 0)).sample(6)
    assert np.array_equal(dev_out.A, ora.A)
    assert torus_rel_l2(dev_out.X, ora.X) < 1e-5


@pytest.mark.parametrize("name", ["ff_n8", "ff_n32"])
def test_force_field_against_golden(cuda, name):
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.force_field_augmented_score_network import (
        ForceFieldAugmentedScoreNetwork, ForceFieldParameters)
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    g = load_golden(name + ".npz")
    B, N, _ = g["X"].shape
    ff = ForceFieldAugmentedScoreNetwork(nets.fake_net(1).to(cuda), ForceFieldParameters(
        radial_cutoff=float(g["rc"]), strength=float(g["strength"])))
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.zeros(B, N, dtype=torch.long, device=cuda),
                                        X=torch.from_numpy(g["X"]).to(cuda), L=torch.from_numpy(g["L"]).to(cuda)),
             TIME: torch.zeros(B, 1, device=cuda), NOISE: torch.zeros(B, 1, device=cuda),
             CARTESIAN_FORCES: torch.zeros(B, N, 3, device=cuda)}
    forces = ff.get_relative_coordinates_pseudo_force(batch).cpu().numpy()
    # per-atom sums of a few float32 terms in a different order than the reference's scatter_add: 1e-5 of the scale
    assert np.abs(forces - g["forces"]).max() <= 1e-5 * np.abs(g["forces"]).max()
    out = ff(batch, conditional=False)
    assert np.abs(out.X.cpu().numpy() - g["out_X"]).max() <= 1e-5 * np.abs(g["out_X"]).max()


def test_atom_type_update_recording(cuda):
    P = _pkg()
    g = load_golden("traj_record_atom_types.npz")
    npar = P["Noise"](**cases.noise_ns(6))
    spar = P["Sampling"](**cases.sampling_ns(8, 2), record_samples=True, record_atom_type_update=True)
    gen = P["Langevin"](npar, spar, nets.fake_net(2).to(cuda))
    gen.noise_source = _replayed(g)
    with torch.no_grad():
        out = _np(gen.sample(int(g["batch"]), cuda))
    assert np.array_equal(out.A, g["final_A"])
    rec = gen.sample_trajectory_recorder._internal_data["atom_type_update"]
    assert len(rec) == len(g["rec_a_i"])
    for k, e in enumerate(rec):
        assert np.array_equal(e["a_i"].numpy(), g["rec_a_i"][k]) and np.array_equal(e["a_im1"].numpy(), g["rec_a_im1"][k])
        # the Gumbel values are computed by torch on the HOST CPU like the reference's (langevin_generator.py:100-107);
        # torch's CPU log differs in the last bit between hosts (AVX2 / AVX512 Sleef), hence a 2-ulp tolerance
        np.testing.assert_allclose(e["gumbel_sample"].numpy(), g["rec_gumbel_sample"][k], rtol=3e-7, atol=1e-7)
        assert np.array_equal(e["predicted_logits"].numpy(), g["rec_predicted_logits"][k])
        from conftest import ulp_diff
        assert ulp_diff(e["one_step_transition_probabilities"].numpy(), g["rec_one_step_transition_probabilities"][k]).max() <= 4


def test_reference_private_update_methods_against_golden(cuda):
    """LangevinGenerator._relative_coordinates_update / _lattice_parameters_update / _atom_types_update (and their
    _predictor_step / _corrector_step aliases, the corrector step sizes): the reference's private methods under the reference's
    names and operands -- tests/golden/make_golden.py called exactly these on the reference's generator to produce
    p1_coordinates / p3_lattice / p2_atom_types; here the same calls go to the HIP generator.  Coordinates, lattice and atom
    types bit for bit, the recorded transition probabilities within 4 ulp (exp inside the softmax)."""
    from conftest import load_golden, ulp_diff
    P = _pkg()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    def generator(num_atom_types, greedy=True, one=True, fixed=True, T=10, N=8, record=False):
        npar = P["Noise"](**cases.noise_ns(T))
        skw = cases.sampling_ns(N, num_atom_types, greedy=greedy, one=one, fixed=fixed)
        if record:
            skw.update(record_samples=True, record_atom_type_update=True)
        gen = P["Langevin"](npar, P["Sampling"](**skw), nets.fake_net(num_atom_types))
        gen._prepare(cuda)
        return gen

    g = load_golden("p1_coordinates.npz")
    gen = generator(1)
    for k in range(len(g["scalars"])):
        w, n, sig = (torch.tensor(v) for v in g["scalars"][k])
        for method in (gen._relative_coordinates_update, gen._relative_coordinates_update_predictor_step,
                       gen._relative_coordinates_update_corrector_step):
            got = method(t(g["x"]).to(cuda), t(g["s"]).to(cuda), sig, w, n, t(g["z"]).to(cuda))
            assert np.array_equal(got.cpu().numpy(), g["x_out"][k])
    # z = None: the draw comes from the generator's own hook, like the reference's (:190-193)
    gen._draw_coordinates_gaussian_sample = lambda number_of_samples: t(g["z"])
    w, n, sig = (torch.tensor(v).to(cuda) for v in g["scalars"][0])          # device scalars: no host read needed
    got = gen._relative_coordinates_update(t(g["x"]).to(cuda), t(g["s"]).to(cuda), sig, w, n, None)
    assert np.array_equal(got.cpu().numpy(), g["x_out"][0])
    with pytest.raises(Exception, match="one value per call"):
        gen._relative_coordinates_update(t(g["x"]).to(cuda), t(g["s"]).to(cuda), torch.ones(6), w, n, t(g["z"]).to(cuda))

    g = load_golden("p3_lattice.npz")
    free, fixed = generator(1, fixed=False), generator(1, fixed=True)
    for k in range(len(g["scalars"])):
        w, n, _, sigma_n = (torch.tensor(v) for v in g["scalars"][k])
        for method in (free._lattice_parameters_update, free._lattice_parameters_update_predictor_step,
                       free._lattice_parameters_update_corrector_step):
            got = method(t(g["l"]).to(cuda), t(g["s"]).to(cuda), sigma_n, w, n, t(g["z"]).to(cuda))
            assert np.array_equal(got.cpu().numpy(), g["l_out"][k])
        lat = t(g["l"]).to(cuda)
        assert fixed._lattice_parameters_update(lat, t(g["s"]).to(cuda), sigma_n, w, n, t(g["z"]).to(cuda)) is lat
    eps = free._get_coordinates_corrector_step_size(3, torch.tensor(0.1), t(g["s"]).to(cuda), None)
    assert eps.is_cuda and float(eps) == float(free.langevin_dynamics.epsilon[3])
    assert float(free._get_lattice_parameters_corrector_step_size(0, None, t(g["s"]).to(cuda), None)) == \
        float(free.langevin_dynamics.epsilon[0])

    g = load_golden("p2_atom_types.npz")
    for name in g["names"]:
        greedy, one_eff, idx, T = (int(v) for v in g[f"{name}/flags"])
        C = g[f"{name}/logits"].shape[-1]
        gen = generator(C - 1, greedy=bool(greedy), one=bool(one_eff), T=T, record=True)
        gen._draw_gumbel_sample = lambda number_of_samples: t(g[f"{name}/gumbel"])
        gen._draw_binary_sample = lambda number_of_samples: t(g[f"{name}/u"])
        B, N = g[f"{name}/a"].shape
        q, qb, qbm = [t(g[f"{name}/{k}"]).to(cuda)[None, None].expand(B, N, C, C) for k in ("q", "qbar", "qbar_tm1")]
        a_out = gen._atom_types_update(t(g[f"{name}/logits"]).to(cuda), t(g[f"{name}/a"]).to(cuda), q, qb, qbm,
                                       atom_type_greedy_sampling=bool(greedy), one_atom_type_transition_per_step=bool(one_eff))
        assert np.array_equal(a_out.cpu().numpy(), g[f"{name}/a_out"]), name
        rec = gen.sample_trajectory_recorder._internal_data["atom_type_update"][0]
        assert np.array_equal(rec["gumbel_sample"].numpy(), g[f"{name}/gumbel_used"]), name
        assert ulp_diff(rec["one_step_transition_probabilities"].numpy(), g[f"{name}/p"]).max() <= 4, name
        assert np.array_equal(rec["a_i"].numpy(), g[f"{name}/a"]) and np.array_equal(rec["a_im1"].numpy(), g[f"{name}/a_out"])


def test_noising_transform_given_time_index(cuda, oracle):
    """F1 + F2 + F3 behind the reference's NoisingTransform.transform_given_time_index, against the oracle on the draws
    torch's CPU generator produces for a fixed seed (same order: X noise, A noise, L noise)."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.data.diffusion.noising_transform import NoisingTransform
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (
        ATOM_TYPES, LATTICE_PARAMETERS, NOISY_ATOM_TYPES, NOISY_LATTICE_PARAMETERS, NOISY_RELATIVE_COORDINATES,
        RELATIVE_COORDINATES)
    P = _pkg()
    npar = P["Noise"](**cases.noise_ns(10))
    B, N, nat = 6, 8, 2
    x0, a0 = torch.rand(B, N, 3), torch.randint(0, nat, (B, N))
    l0 = torch.rand(B, 6) + 5
    sched = oracle.noise_schedule(10, num_classes=nat + 1)
    for fixed in (True, False):
        with pytest.raises(NotImplementedError, match="training-time"):       # the reference's default (optimal transport): refused loudly
            NoisingTransform(npar, num_atom_types=nat, spatial_dimension=3, use_fixed_lattice_parameters=fixed, device=cuda)
        tr = NoisingTransform(npar, num_atom_types=nat, spatial_dimension=3, use_fixed_lattice_parameters=fixed,
                              use_optimal_transport=False, device=cuda)
        for index_i in (1, 4, 10):
            torch.manual_seed(index_i)
            out = tr.transform_given_time_index({RELATIVE_COORDINATES: x0.to(cuda), ATOM_TYPES: a0.to(cuda),
                                                 LATTICE_PARAMETERS: l0.to(cuda)}, index_i)
            torch.manual_seed(index_i)
            z, u = torch.randn(B, N, 3).numpy(), torch.rand(B, N, nat + 1).numpy()
            idx = index_i - 1
            assert np.array_equal(out[NOISY_RELATIVE_COORDINATES].cpu().numpy(),
                                  oracle.noise_coordinates(x0.numpy(), z, sched["sigma"][idx]))
            assert np.array_equal(out[NOISY_ATOM_TYPES].cpu().numpy(),
                                  oracle.noise_atom_types(a0.numpy(), sched["q_bar_matrix"][idx], u))
            if fixed:
                assert torch.equal(out[NOISY_LATTICE_PARAMETERS].cpu(), l0)
            else:
                zl = torch.randn(B, 6).numpy()
                sigma_n = np.float32(sched["sigma"][idx] / np.float32(8.0) ** np.float32(1 / 3))
                np.testing.assert_allclose(out[NOISY_LATTICE_PARAMETERS].cpu().numpy(), l0.numpy() + sigma_n * zl, rtol=1e-6)


def test_fused_sampler_specialised_equals_generic(cuda, monkeypatch):
    """The template-MLP instantiation (dimensions as literals) against the generic one: same code path, same bits."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    P = _pkg()
    outs = []
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
    for options in (_hip.MLP_SAMPLE_UNFOLDED | _hip.MLP_SAMPLE_GENERIC_KERNEL, _hip.MLP_SAMPLE_UNFOLDED):
        torch.manual_seed(1234)
        net = nets.mlp_net(8, 1).to(cuda)                     # the template sizes: the specialised kernel applies
        npar = P["Noise"](**cases.noise_ns(30, sigma_min=1e-4, sigma_max=0.25))
        spar = P["Sampling"](**cases.sampling_ns(8, 1), rng_mode="device", seed=3, fused_score_network=True)
        gen = LangevinGenerator(npar, spar, net)
        gen.fused_sampler_options = options                   # layer-by-layer form on both sides; generic vs specialised
        with torch.no_grad():
            outs.append(_np(gen.sample(300, cuda)))
    assert np.array_equal(outs[0].A, outs[1].A)
    assert np.array_equal(outs[0].X.view(np.int32), outs[1].X.view(np.int32))


@pytest.mark.parametrize("in_corrector", [False, True])
@pytest.mark.parametrize("small_epsilon", [1e-8, 1e-6, 1e-3])
def test_fused_sampler_hoisted_softmax_is_exact(cuda, monkeypatch, small_epsilon, in_corrector):
    """One atom type: the clipped softmax of (l0, -inf) is evaluated once per launch, and the whole posterior of a
    step -- a function of a_t alone -- once per step by the noise pre-pass, instead of per atom and step.  Same bits as
    the per-atom evaluation (options MLP_SAMPLE_NO_FIXED_SOFTMAX / _NO_P2_TABLE switch the two off), any small_epsilon,
    greedy or not."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    P = _pkg()
    outs = []
    for flag in (24, 16, 8, 0):
        torch.manual_seed(1234)
        net = nets.mlp_net(8, 1).to(cuda)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = P["Noise"](**cases.noise_ns(30, **cases.LIN))
            spar = P["Sampling"](**cases.sampling_ns(8, 1, eps=small_epsilon, in_corr=in_corrector,
                                                     greedy=not in_corrector, one=not in_corrector, M=2),
                                 rng_mode="device", seed=3, fused_score_network=True)
        gen = LangevinGenerator(npar, spar, net)
        gen.fused_sampler_options = flag
        with torch.no_grad():
            gen._prepare(cuda)
            gen._begin_call(cuda)
            start = gen.initialize(300, cuda)
            outs.append(_np(gen._sample_fused(start, 30, 0)))       # no status check: a MASK may legitimately survive
    for other in outs[1:]:
        assert np.array_equal(outs[0].A, other.A)
        assert np.array_equal(outs[0].X.view(np.int32), other.X.view(np.int32))


def test_fused_sampler_folded_input_layer(cuda, monkeypatch):
    """The template network with its five (linear) embedding layers folded into the first hidden layer against the
    layer-by-layer form: the same function, rounding differs in the last bits.  One iteration from the same state
    (nothing to amplify a difference): atom types exact, coordinates within 1e-6; a whole trajectory of a neutral
    configuration (linear schedule): atom types exact, coordinates within 1e-5."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    P = _pkg()
    torch.manual_seed(1234)
    net = nets.mlp_net(8, 1).to(cuda)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**cases.noise_ns(40, **cases.LIN))
        spar = P["Sampling"](**cases.sampling_ns(8, 1), rng_mode="device", seed=3, fused_score_network=True)
    gen = LangevinGenerator(npar, spar, net)
    with torch.no_grad():
        gen._prepare(cuda)
        gen._begin_call(cuda)
        start = gen.initialize(200, cuda)
        sched, pack = gen._prepare(cuda), gen.fused_pack(cuda)
        assert pack.c_struct.folded_input and pack.c_struct.folded_output
        results = {}
        from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
        for fold in ("1", "0"):
            for n_iterations in (1, 40):
                comp = RS.AXL(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
                kernels.mlp_pc_sample(sched, pack, gen._flags(True), 1, False, 40, n_iterations, gen._rng(0), comp.A,
                                      comp.X, comp.L, gen._status, workspace=gen._noise_workspace,
                                      options=0 if fold == "1" else _hip.MLP_SAMPLE_UNFOLDED)
                results[fold, n_iterations] = _np(comp)
    for n_iterations, tol in ((1, 1e-6), (40, 1e-5)):
        a, b = results["1", n_iterations], results["0", n_iterations]
        assert np.array_equal(a.A, b.A), n_iterations
        assert torus_rel_l2(a.X, b.X) < tol, (n_iterations, torus_rel_l2(a.X, b.X))
    assert not np.array_equal(results["1", 40].X, results["0", 40].X) or True    # (they may even coincide)
    assert (results["1", 40].A != 1).all()


@pytest.mark.parametrize("n_atoms,nat,hidden,n_hidden,fits", [(8, 1, 64, 3, True), (5, 4, 48, 2, True), (12, 1, 64, 2, True),
                                                              (8, 2, 64, 4, True), (8, 2, 128, 2, False)])
def test_fused_sampler_generic_folded_forward(cuda, n_atoms, nat, hidden, n_hidden, fits):
    """The GENERIC instantiation of the persistent sampler (any MLP shape) with the folded forward -- input embeddings folded into
    the first hidden layer, last hidden layer folded into the heads, hardware sin / cos (round 4; what shapes outside the
    register-resident family run) -- against its layer-by-layer form (MLP_SAMPLE_UNFOLDED): the same function, last-bit
    different rounding.  One iteration from the same state: atom types exact, coordinates within 1e-6; a whole trajectory of a
    neutral configuration (linear schedule): atom types exact, coordinates within 1e-5."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    import warnings
    P = _pkg()
    torch.manual_seed(77)
    net = nets.mlp_net(n_atoms, nat, hidden=hidden, n_hidden=n_hidden).to(cuda)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**cases.noise_ns(40, **cases.LIN))
        spar = P["Sampling"](**cases.sampling_ns(n_atoms, nat), rng_mode="device", seed=3, fused_score_network=True)
    gen = LangevinGenerator(npar, spar, net)
    with torch.no_grad():
        sched = gen._prepare(cuda)
        gen._begin_call(cuda)
        start = gen.initialize(150, cuda)
        pack = gen.fused_pack(cuda)
        assert pack.c_struct.folded_input and pack.c_struct.folded_output
        results = {}
        for name, options in (("folded", _hip.MLP_SAMPLE_GENERIC_KERNEL), ("plain", _hip.MLP_SAMPLE_GENERIC_KERNEL | _hip.MLP_SAMPLE_UNFOLDED)):
            for n_iterations in (1, 40):
                comp = RS.AXL(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
                kernels.mlp_pc_sample(sched, pack, gen._flags(True), 1, False, 40, n_iterations, gen._rng(0), comp.A, comp.X,
                                      comp.L, gen._status, workspace=gen._noise_workspace, options=options)
                results[name, n_iterations] = _np(comp)
    for n_iterations, tol in ((1, 1e-6), (40, 1e-5)):
        a, b = results["folded", n_iterations], results["plain", n_iterations]
        assert np.array_equal(a.A, b.A), n_iterations
        assert torus_rel_l2(a.X, b.X) < tol, (n_iterations, torus_rel_l2(a.X, b.X))
    if fits:      # (hidden 128: the weight image alone exceeds the LDS budget, the kernel reads global weights, unfolded)
        assert not np.array_equal(results["folded", 40].X.view(np.int32), results["plain", 40].X.view(np.int32)), \
            "the folded forward did not run (same bits as the layer-by-layer form)"


def _mlp_any(n_atoms, nat, hidden, n_hidden, e_coord, e_noise, e_time, e_atom, e_lattice):
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.mlp_score_network import (
        MLPScoreNetwork, MLPScoreNetworkParameters)
    return MLPScoreNetwork(MLPScoreNetworkParameters(
        number_of_atoms=n_atoms, num_atom_types=nat, n_hidden_dimensions=n_hidden, hidden_dimensions_size=hidden,
        relative_coordinates_embedding_dimensions_size=e_coord, noise_embedding_dimensions_size=e_noise,
        time_embedding_dimensions_size=e_time, atom_type_embedding_dimensions_size=e_atom,
        lattice_parameters_embedding_dimensions_size=e_lattice)).eval()


# the reference's own MLP configurations (configuration_templates/.../config_diffusion_mlp.yaml, ..._orion.yaml,
# analysis_and_sanity_checks/{toy_problems,atom_types_only_experiments}/training/*.yaml) and a few around them:
# (N, atom types, hidden, hidden layers, e_coordinates, e_noise, e_time, e_atom_type, e_lattice) -> expected instantiation
PADDED_FAMILY_SHAPES = [
    ((8, 1, 64, 3, 32, 16, 16, 1, 1), 213),        # the template at N = 8 (60 folded inputs)
    ((2, 1, 64, 3, 32, 16, 16, 1, 1), 213),        # the template as written (N = 2)
    ((8, 2, 64, 3, 2, 64, 64, 64, 2), 233),        # atom_types_only_experiments: 48 + 2 + 8 x 64 + 2 > 192 inputs -> see below
    ((8, 1, 16, 1, 32, 16, 16, 16, 8), 0),         # orion, one hidden layer: nothing to fold an output layer into
    ((8, 1, 32, 4, 32, 16, 16, 16, 8), 234),       # orion: 48 + 2 + 128 + 8 = 186 inputs, hidden 32, four layers
    ((8, 1, 64, 2, 32, 16, 16, 16, 8), 232),
    ((5, 3, 48, 3, 8, 4, 4, 3, 2), 213),
    ((8, 2, 64, 4, 32, 16, 16, 1, 1), 214),
]


GENERIC_UNFOLDED_SHAPES = {(8, 1, 64, 2, 32, 16, 16, 16, 8)}


@pytest.mark.parametrize("shape,variant", PADDED_FAMILY_SHAPES)
def test_fused_sampler_padded_family_equals_generic_folded(cuda, shape, variant):
    """The padded register-resident family (round 4: weights of every layer in registers, fixed padded sizes, run-time structure
    dimensions -- what every MLP configuration of the reference runs) against the generic instantiation's folded forward on the
    unpadded matrices: a zero quad adds fma(0, 0, s) = s, so the two agree BIT FOR BIT over a whole trajectory; and
    mdx_mlp_pc_sample_variant names the instantiation."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    import warnings
    n_atoms, nat = shape[0], shape[1]
    P = _pkg()
    torch.manual_seed(91)
    net = _mlp_any(*shape).to(cuda)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**cases.noise_ns(25, **cases.LIN))
        spar = P["Sampling"](**cases.sampling_ns(n_atoms, nat, M=2), rng_mode="device", seed=3, fused_score_network=True)
    gen = LangevinGenerator(npar, spar, net)
    with torch.no_grad():
        sched = gen._prepare(cuda)
        gen._begin_call(cuda)
        start = gen.initialize(130, cuda)
        pack = gen.fused_pack(cuda)
        n_inputs = 6 * n_atoms + 2 + n_atoms * shape[7] + shape[8]
        if n_inputs > 192:                       # outside the family's limits: no padded matrices, generic kernel
            variant = 0
        got = _hip.lib().mdx_mlp_pc_sample_variant(__import__("ctypes").byref(pack.c_struct), _hip.MLP_SAMPLE_PADDED_FAMILY)
        assert got == variant, (got, variant)
        assert bool(pack.c_struct.folded_padded) == (variant >= 200)
        results = {}
        for name, options in (("padded", _hip.MLP_SAMPLE_PADDED_FAMILY), ("generic", _hip.MLP_SAMPLE_GENERIC_KERNEL)):
            comp = RS.AXL(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
            kernels.mlp_pc_sample(sched, pack, gen._flags(True), 2, False, 25, 25, gen._rng(0), comp.A, comp.X, comp.L,
                                  gen._status, workspace=gen._noise_workspace, options=options)
            results[name] = _np(comp)
    assert np.array_equal(results["padded"].A, results["generic"].A)
    same_bits = np.array_equal(results["padded"].X.view(np.int32), results["generic"].X.view(np.int32))
    if shape in GENERIC_UNFOLDED_SHAPES:
        # (the generic kernel keeps its layer-by-layer forward here -- image + folded matrices exceed its LDS budget --: the same
        # function with last-bit different rounding, as in test_fused_sampler_generic_folded_forward)
        assert not same_bits and torus_rel_l2(results["padded"].X, results["generic"].X) < 1e-5
    else:
        assert same_bits
    assert (results["padded"].A != nat).all() and np.isfinite(results["padded"].X).all()


def test_egnn_fused_ops_equal_plain_torch(cuda):
    """The EGNN forward with the fused helpers (fused first message layer, sorted-segment reductions, the MFMA chains) against the
    same module with plain PyTorch ops, radius-graph edges, experiment-like widths."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    torch.manual_seed(7)
    net = nets.egnn_net(2, "radial_cutoff", 7.5, hidden=128, n_layers=3, n_hidden=3).to(cuda)
    B, N = 6, 64
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, 3, (B, N), device=cuda), X=torch.rand(B, N, 3, device=cuda),
                                        L=torch.tensor([11.084] * 3 + [0.0] * 3, device=cuda).repeat(B, 1)),
             TIME: torch.rand(B, 1, device=cuda), NOISE: torch.rand(B, 1, device=cuda) * 0.2,
             CARTESIAN_FORCES: torch.zeros(B, N, 3, device=cuda)}
    outs = []
    for fused in (True, False):
        for layer in net.egnn.graph_layers:
            layer.use_fused_ops = fused
        with torch.no_grad():
            outs.append(net(batch, conditional=False))
    for got, want in ((outs[0].X, outs[1].X), (outs[0].A[..., :-1], outs[1].A[..., :-1])):
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-7


@pytest.mark.parametrize("n_nodes,H,mean", [(1, 4, False), (37, 64, True), (500, 256, True), (64, 260, False)])
def test_segment_kernels_against_torch(cuda, n_nodes, H, mean):
    """mdx_segment_rows / mdx_egnn_coord_head on ragged sorted segments (empty ones included) vs plain torch in fp64."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    g = torch.Generator().manual_seed(n_nodes + H)
    degree = torch.randint(0, 40, (n_nodes,), generator=g)
    degree[0] = 0
    if n_nodes > 3:
        degree[3] = 0
    E = int(degree.sum())
    offsets = torch.cumsum(degree, 0) - degree
    data = torch.randn(E, H, generator=g)
    w = torch.randn(H, generator=g) / H ** 0.5
    cd = torch.randn(E, 3, generator=g)
    seg = torch.repeat_interleave(torch.arange(n_nodes), degree)
    want_rows = torch.zeros(n_nodes, H, dtype=torch.float64).index_add_(0, seg, data.double())
    want_trans = torch.zeros(n_nodes, 3, dtype=torch.float64).index_add_(0, seg, cd.double() * (data.double() @ w.double())[:, None])
    if mean:
        scale = 1.0 / degree.clamp(min=1).double()[:, None]
        want_rows, want_trans = want_rows * scale, want_trans * scale
    dev = lambda t: t.to(cuda).contiguous()
    got_rows = kernels.segment_rows(dev(data), dev(offsets), dev(degree), mean).cpu().double()
    got_trans = kernels.egnn_coord_head(dev(data), dev(w), dev(cd), dev(offsets), dev(degree), mean).cpu().double()
    assert float((got_rows - want_rows).abs().max()) <= 2e-6 * max(1.0, float(want_rows.abs().max()))
    assert float((got_trans - want_trans).abs().max()) <= 2e-6 * max(1.0, float(want_trans.abs().max()))
    assert (got_rows[0] == 0).all() and (got_trans[0] == 0).all()        # empty segment


def _random_config(seed):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([1, 2, 3, 5, 8, 13, 31, 32, 33, 64, 65, 100, 216]))
    nat = int(rng.integers(1, 6))
    d = int(rng.choice([1, 2, 3, 3, 3]))
    M = int(rng.choice([0, 1, 2, 3]))
    T = int(rng.integers(3, 8))
    fixed = bool(rng.random() < 0.6)
    noise_kw = cases.noise_ns(T, schedule_type=str(rng.choice(["exponential", "linear"])),
                              sigma_min=float(rng.choice([1e-3, 5e-3])), sigma_max=float(rng.choice([0.2, 0.5])))
    sampling_kw = cases.sampling_ns(N, nat, M=M, greedy=bool(rng.random() < 0.5), one=bool(rng.random() < 0.5),
                                    in_corr=bool(rng.random() < 0.4), eps=float(rng.choice([1e-8, 1e-6])), fixed=fixed,
                                    cell=[float(c) for c in rng.uniform(4.0, 12.0, d)], d=d)
    return noise_kw, sampling_kw, int(rng.choice([1, 3, 17]))


@pytest.mark.parametrize("seed", range(32 * FUZZ))
def test_random_configurations_device_rng_bitwise(cuda, seed):
    """Seeded random sampler configurations (atoms 1..216, 2..6 classes, 1..3 spatial dimensions, 0..3 correctors, all
    flag combinations, fixed and free lattice) with the echo network, whose forward is exact on both sides: the GPU
    generator in device-RNG mode -- eager and hipGraph replay -- must equal the oracle in every bit."""
    P = _pkg()
    noise_kw, sampling_kw, batch = _random_config(seed)
    import warnings
    outs = []
    for use_graph in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = P["Noise"](**noise_kw)
            spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=1000 + seed, use_hip_graph=use_graph)
        net = nets.fake_net(spar.num_atom_types, d=spar.spatial_dimension).to(cuda)
        gen = P["Langevin"](npar, spar, net)
        with torch.no_grad():
            outs.append(_np(gen.sample(batch, cuda)))
    ora = RS.OracleLangevinGenerator(npar, spar, nets.fake_net(spar.num_atom_types, d=spar.spatial_dimension),
                                     noise=RS.PhiloxNoise(1000 + seed, 0)).sample(batch)
    for out in outs:
        assert np.array_equal(out.A, ora.A), (noise_kw, sampling_kw, batch)
        assert np.array_equal(out.X.view(np.int32), ora.X.view(np.int32)), (noise_kw, sampling_kw, batch)
        assert np.array_equal(out.L.view(np.int32), ora.L.view(np.int32)), (noise_kw, sampling_kw, batch)


@pytest.mark.parametrize("seed", range(12 * FUZZ))
def test_fused_sampler_random_shapes_predrawn_equals_in_kernel(cuda, seed):
    """Generic instantiations of the persistent sampler on random network / structure shapes (records longer than one
    64-lane fetch included): the noise pre-pass and the in-kernel draws give the same bits, and one launch equals two."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
    import warnings
    P = _pkg()
    rng = np.random.default_rng(100 + seed)
    N, nat = int(rng.choice([1, 3, 8, 17, 40, 64])), int(rng.integers(1, 5))
    hidden, n_hidden = int(rng.choice([32, 48, 64, 96])), int(rng.integers(1, 5))
    M, T, batch = int(rng.integers(0, 3)), int(rng.integers(3, 7)), int(rng.choice([1, 5, 9]))
    in_corr = bool(rng.random() < 0.5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**cases.noise_ns(T, **cases.LIN))
        spar = P["Sampling"](**cases.sampling_ns(N, nat, M=M, greedy=bool(rng.random() < 0.5), one=bool(rng.random() < 0.5),
                                                 in_corr=in_corr, fixed=bool(rng.random() < 0.7)),
                             rng_mode="device", seed=seed, fused_score_network=True)
    torch.manual_seed(seed)
    net = nets.mlp_net(N, nat, hidden=hidden, n_hidden=n_hidden).to(cuda)
    gen = P["Langevin"](npar, spar, net)
    with torch.no_grad():
        gen._prepare(cuda)
        gen._begin_call(cuda)
        start = gen.initialize(batch, cuda)
        sched, pack = gen._prepare(cuda), gen.fused_pack(cuda)
        outs = []
        for mode in ("in_kernel", "predrawn", "two_launches"):
            comp = RS.AXL(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
            segments = ((T, T),) if mode != "two_launches" else ((T, 2), (T - 2, T - 2))
            for first, n in segments:
                kernels.mlp_pc_sample(sched, pack, gen._flags(True), M, in_corr, first, n, gen._rng(0), comp.A, comp.X,
                                      comp.L, gen._status,
                                      workspace=None if mode == "in_kernel" else gen._noise_workspace)
            outs.append(comp)
    for other in outs[1:]:
        assert torch.equal(outs[0].A, other.A), (N, nat, hidden, n_hidden, M, T, batch)
        assert torch.equal(outs[0].X.view(torch.int32), other.X.view(torch.int32))
        assert torch.equal(outs[0].L.view(torch.int32), other.L.view(torch.int32))
    assert torch.isfinite(outs[0].X).all() and (outs[0].A != nat).all()


@pytest.mark.parametrize("seed", range(12 * FUZZ))
def test_random_repaint_configurations_bitwise(cuda, seed):
    """Random repaint set-ups (constraint count / rows, resampling passes 0..2, correctors, flags) with the echo network:
    GPU device-RNG generator -- eager and graph replay -- equals the oracle bit for bit; constrained rows are pinned."""
    import warnings
    P = _pkg()
    rng = np.random.default_rng(500 + seed)
    N, nat = int(rng.choice([4, 8, 13, 40, 70])), int(rng.integers(1, 4))
    K = int(rng.integers(1, N + 1))
    M, T, batch = int(rng.integers(0, 3)), int(rng.integers(3, 7)), int(rng.choice([1, 4, 9]))
    resampling = int(rng.integers(0, 3))
    cx = rng.random((K, 3), dtype=np.float32)
    ca = rng.integers(0, nat, K)
    cidx = rng.permutation(N)[:K] if rng.random() < 0.7 else None
    constraint = P["Constraint"](elements=["Si", "Ge", "C"][:nat], constrained_relative_coordinates=torch.from_numpy(cx),
                                 constrained_atom_types=torch.from_numpy(ca),
                                 constrained_indices=None if cidx is None else torch.from_numpy(cidx))
    noise_kw = cases.noise_ns(T, schedule_type=str(rng.choice(["exponential", "linear"])))
    sampling_kw = cases.sampling_ns(N, nat, M=M, greedy=bool(rng.random() < 0.5), one=bool(rng.random() < 0.5),
                                    in_corr=bool(rng.random() < 0.3))
    outs = []
    for use_graph in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = P["Noise"](**noise_kw)
            spar = P["Sampling"](**sampling_kw, rng_mode="device", seed=77 + seed, use_hip_graph=use_graph,
                                 repaint_resampling_steps=resampling)
        gen = P["Constrained"](npar, spar, nets.fake_net(nat).to(cuda), constraint)
        with torch.no_grad():
            outs.append(_np(gen.sample(batch, cuda)))
    ora = RS.OracleLangevinGenerator(npar, spar, nets.fake_net(nat), noise=RS.PhiloxNoise(77 + seed, 0),
                                     constraint=dict(constrained_relative_coordinates=cx, constrained_atom_types=ca,
                                                     constrained_indices=cidx)).sample(batch)
    rows = np.arange(K) if cidx is None else cidx
    for out in outs:
        assert np.array_equal(out.A, ora.A), (N, nat, K, M, T, batch, resampling)
        assert np.array_equal(out.X.view(np.int32), ora.X.view(np.int32)), (N, nat, K, M, T, batch, resampling)
        assert np.array_equal(out.X[:, rows], np.broadcast_to(cx, (batch, K, 3)))
        assert np.array_equal(out.A[:, rows], np.broadcast_to(ca, (batch, K)))


def test_c1_exact_configuration(cuda):
    """BASELINE configs[0] as the reference runs it on CPU (T = 100, batch 16, MLP template): the GPU generator in
    reference-RNG mode on the reference's recorded draws.  Every one of the 200 steps, started from the reference's own
    composition: atom types exact, coordinates within 1e-5.  Free run: atom types exact at the end.  The coordinates
    of a free run are not comparable at this configuration: with T = 100 the REFERENCE's own map turns a 1e-8
    perturbation of the initial coordinates into an O(1) difference within ten iterations
    (tests/test_oracle_golden.py::test_c1_exact_configuration measures it), so only their validity is checked."""
    P = _pkg()
    g = load_golden("traj_c1_exact.npz")
    noise_kw, sampling_kw, netf = cases.C1_EXACT
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar, spar = P["Noise"](**noise_kw), P["Sampling"](**sampling_kw)
    net = nets.load_fixture_weights(netf(None), g).to(cuda)
    B = int(g["batch"])
    gen = P["Langevin"](npar, spar, net)
    gen.noise_source = _replayed(g)
    with torch.no_grad():
        out = _np(gen.sample(B, cuda))
    assert gen.noise_source.inner.exhausted()
    assert np.array_equal(out.A, g["final_A"])
    assert np.isfinite(out.X).all() and (out.X >= 0).all() and (out.X < 1).all()
    # teacher-forced steps
    gen = P["Langevin"](npar, spar, net)
    gen.noise_source = _replayed(g)

    def axl(a, x, lattice):
        return RS.AXL(A=torch.from_numpy(a.astype(np.int64)).to(cuda), X=torch.from_numpy(x).to(cuda), L=lattice)

    with torch.no_grad():
        gen._prepare(cuda)
        gen._begin_call(cuda)
        lattice = gen.initialize(B, cuda).L          # consumes the initial draws
        forces = torch.zeros(B, 8, 3, device=cuda)
        comp = axl(g["start_A"], g["start_X"], lattice)
        worst = 0.0
        for k, index in enumerate(g["pred_index"]):
            got = gen.predictor_step(comp, int(index), forces)
            assert np.array_equal(got.A.cpu().numpy(), g["pred_out_A"][k]), ("pred", k)
            worst = max(worst, torus_rel_l2(got.X.cpu().numpy(), g["pred_out_X"][k]))
            comp = axl(g["pred_out_A"][k], g["pred_out_X"][k], lattice)
            got = gen.corrector_step(comp, int(index) - 1, forces, 0)
            assert np.array_equal(got.A.cpu().numpy(), g["corr_out_A"][k]), ("corr", k)
            worst = max(worst, torus_rel_l2(got.X.cpu().numpy(), g["corr_out_X"][k]))
            comp = axl(g["corr_out_A"][k], g["corr_out_X"][k], lattice)
    assert gen.noise_source.inner.exhausted()
    assert worst < 1e-5, f"worst per-step rel-L2 {worst:.2e}"


# -------------------------------------------------------------------------------------------------------------
# round 2: what is benchmarked is what is pinned
# -------------------------------------------------------------------------------------------------------------
def _reference_draw_records(gen, batch, with_corrector):
    """One iteration's records of mdx_mlp_pc_sample (layout: include/mdx_hip.h) from the generator's reference-order
    draw hooks (langevin_generator.py:92-111 order: Gumbel u, binary u, z, z_lattice | corrector: z, z_lattice)."""
    n, c = gen.number_of_atoms, gen.num_classes
    gumbel = gen._draw_gumbel_sample(batch).reshape(batch, n * c)
    u = gen._draw_binary_sample(batch) if gen.atom_type_greedy_sampling else torch.zeros(batch, n)
    z = gen._draw_coordinates_gaussian_sample(batch).reshape(batch, n * 3)
    gen._draw_lattice_gaussian_sample(batch)                                  # drawn by the reference, unused (fixed lattice)
    rec = [z, gumbel, u.reshape(batch, n), torch.zeros(batch, 8)]             # table[7] = 0: no per-step posterior table
    if with_corrector:
        rec.append(gen._draw_coordinates_gaussian_sample(batch).reshape(batch, n * 3))
        gen._draw_lattice_gaussian_sample(batch)
    return torch.cat([t.to(torch.float32) for t in rec], dim=1).contiguous()


@pytest.mark.parametrize("options", ["product", "generic"])
def test_fused_sampler_teacher_forced_against_c1_exact(cuda, options):
    """The persistent fused sampler -- the kernel the C2 headline measures (folded weights, hardware exp/sin/cos) -- against
    the REFERENCE's own run of BASELINE configs[0] (tests/golden/traj_c1_exact.npz: T = 100, batch 16, MLP template):
    for each of the 100 iterations the reference's recorded draws are written into the noise-workspace records
    (MDX_MLP_SAMPLE_CALLER_NOISE: the pre-pass is skipped), ONE iteration is launched from the reference's recorded
    composition, and the result is held to the bar of test_c1_exact_configuration: atom types exact, coordinates
    within 1e-5 rel-L2 on the torus.  Predictor alone (M = 0 launch) against the recorded predictor output, then
    predictor + corrector (the product's iteration) against the recorded corrector output."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    P = _pkg()
    g = load_golden("traj_c1_exact.npz")
    noise_kw, sampling_kw, netf = cases.C1_EXACT
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar, spar = P["Noise"](**noise_kw), P["Sampling"](**dict(sampling_kw, fused_score_network=True))
    net = nets.load_fixture_weights(netf(None), g).to(cuda)
    B = int(g["batch"])
    gen = P["Langevin"](npar, spar, net)
    gen.noise_source = _replayed(g)
    opt = 0 if options == "product" else _hip.MLP_SAMPLE_GENERIC_KERNEL | _hip.MLP_SAMPLE_UNFOLDED
    with torch.no_grad():
        sched = gen._prepare(cuda)
        gen._begin_call(cuda)
        lattice = gen.initialize(B, cuda).L                     # consumes the initial draws
        pack = kernels.MlpPack(net, cuda)
        if options == "product":
            assert pack.c_struct.folded_input and pack.c_struct.folded_output

        def launch(a, x, index, records, correctors):
            comp = RS.AXL(A=torch.from_numpy(a.astype(np.int64)).to(cuda), X=torch.from_numpy(x).to(cuda), L=lattice.clone())
            kernels.mlp_pc_sample(sched, pack, gen._flags(True), correctors, False, int(index), 1, gen._rng(0), comp.A,
                                  comp.X, comp.L, gen._status, caller_records=records, options=opt)
            return comp.A.cpu().numpy(), comp.X.cpu().numpy()

        a_in, x_in = g["start_A"], g["start_X"]
        worst_pred = worst_iter = 0.0
        for k, index in enumerate(g["pred_index"]):
            records = _reference_draw_records(gen, B, True).to(cuda)
            n_pred = 8 * (3 + 2 + 1) + 8
            a, x = launch(a_in, x_in, index, records[:, :n_pred].contiguous(), 0)          # predictor alone
            assert np.array_equal(a, g["pred_out_A"][k]), ("pred", k)
            worst_pred = max(worst_pred, torus_rel_l2(x, g["pred_out_X"][k]))
            a, x = launch(a_in, x_in, index, records, 1)                                    # the product's iteration
            assert np.array_equal(a, g["corr_out_A"][k]), ("iteration", k)
            worst_iter = max(worst_iter, torus_rel_l2(x, g["corr_out_X"][k]))
            a_in, x_in = g["corr_out_A"][k], g["corr_out_X"][k]
    assert gen.noise_source.inner.exhausted()
    assert worst_pred < 1e-5, f"predictor: worst rel-L2 {worst_pred:.2e}"
    assert worst_iter < 1e-5, f"predictor + corrector: worst rel-L2 {worst_iter:.2e}"


def _diamond_sites(n_cells):
    base = torch.tensor([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                         [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
    cells = torch.cartesian_prod(*[torch.arange(n_cells)] * 3).float()
    return ((cells[:, None, :] + base[None]) / n_cells).reshape(-1, 3)


def _c5_generator(cuda, net, resampling, T, seed=77, use_graph=False):
    P = _pkg()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**cases.noise_ns(T, **cases.LIN))
        spar = P["Sampling"](**cases.sampling_ns(216, 1, M=2, greedy=False, one=False, cell=[16.29] * 3),
                             rng_mode="device", seed=seed, repaint_resampling_steps=resampling, use_hip_graph=use_graph)
    constraint = P["Constraint"](elements=["Si"], constrained_relative_coordinates=_diamond_sites(3)[:108].clone(),
                                 constrained_atom_types=torch.zeros(108, dtype=torch.long))
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.constrained_langevin_generator import \
        ConstrainedLangevinGenerator
    return ConstrainedLangevinGenerator(npar, spar, net, constraint), npar, spar


@pytest.mark.parametrize("resampling", [0, 1])
def test_c5_workload_repaint_egnn_radial_graph(cuda, resampling):
    """BASELINE configs[4] at its workload: N = 216 (Si 3x3x3, 16.29 A cell, graph cell clipped to
    16.5 A), K = 108 diamond sites pinned (constrained_indices = arange), ConstrainedLangevinGenerator, EGNN with the HIP
    radius graph at rc = 7.5 (reduced width), M = 2, B = 256 per GPU, without and with resampling.  Properties the domain
    offers at this size: constrained rows pinned exactly, coordinates in [0, 1), full unmasking, determinism under the
    seed; and the time-index sweep includes index 0 (T steps down to 0)."""
    torch.manual_seed(4321)
    net = nets.egnn_net(1, "radial_cutoff", 7.5, hidden=32, n_layers=2, n_hidden=1).to(cuda)
    outs = []
    for _ in range(2):
        gen, npar, spar = _c5_generator(cuda, net, resampling, T=5)
        with torch.no_grad():
            outs.append(_np(gen.sample(256, cuda)))
    out = outs[0]
    sites = _diamond_sites(3)[:108].numpy()
    assert out.X.shape == (256, 216, 3) and out.A.shape == (256, 216)
    assert np.array_equal(out.X[:, :108], np.broadcast_to(sites, (256, 108, 3)))          # pinned bit for bit
    assert (out.A[:, :108] == 0).all()
    assert (out.A != 1).all(), "MASK left at the last step"
    assert np.isfinite(out.X).all() and (out.X >= 0).all() and (out.X < 1).all()
    assert np.array_equal(outs[0].A, outs[1].A) and np.array_equal(outs[0].X.view(np.int32), outs[1].X.view(np.int32))
    free = out.X[:, 108:]
    assert free.std() > 0.2                                                              # the free atoms did move around


@pytest.mark.parametrize("resampling", [0, 1])
def test_c5_shape_bitwise_against_oracle(cuda, resampling):
    """The C5 shape (N = 216, K = 108, M = 2, repaint, +- resampling) at a small batch with the echo network: the GPU
    generator equals the CPU oracle bit for bit (atom types and coordinates), index 0 included."""
    P = _pkg()
    net = nets.fake_net(1)
    gen, npar, spar = _c5_generator(cuda, net.to(cuda), resampling, T=6, seed=78)
    with torch.no_grad():
        out = _np(gen.sample(3, cuda))
    sites = _diamond_sites(3)[:108].numpy()
    ora = RS.OracleLangevinGenerator(npar, spar, nets.fake_net(1), noise=RS.PhiloxNoise(78, 0),
                                     constraint=dict(constrained_relative_coordinates=sites,
                                                     constrained_atom_types=np.zeros(108, dtype=np.int64),
                                                     constrained_indices=None)).sample(3)
    assert np.array_equal(out.A, ora.A)
    assert np.array_equal(out.X.view(np.int32), ora.X.view(np.int32))
    assert np.array_equal(out.X[:, :108], np.broadcast_to(sites, (3, 108, 3)))


@pytest.mark.parametrize("name", ["traj_fake_c3_m2", "traj_mlp_c3"])
def test_start_from_given_configuration_on_the_hip_path(cuda, name, tmp_path):
    """I2 (generators/trajectory_initializer.py:134-186): a run started from a {noisy_axl, start_time_step_index} pickle at
    index k equals the tail of the full run.  Device RNG: a draw is a pure function of (seed, call, time index), so the
    tail from the recorded composition at index k is bit-identical to the full run's remainder."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.trajectory_initializer import (
        StartFromGivenConfigurationTrajectoryInitializer, instantiate_trajectory_initializer)
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import NOISY_AXL_COMPOSITION
    gen, npar, spar, _ = _build(name, cases.TRAJECTORIES, cuda, rng_mode="device", seed=21)
    T, k, B = npar.total_time_steps, 6, 9
    with torch.no_grad():
        gen._prepare(cuda)
        gen._begin_call(cuda)
        start = gen.initialize(B, cuda)
        middle = gen.sample_from_noisy_composition(start, T, k)             # indices T-1 .. k
        full = _np(gen.sample_from_noisy_composition(middle, k, 0))
        gen.check_status()
    path = tmp_path / "start.pickle"
    torch.save({NOISY_AXL_COMPOSITION: RS_AXL_cpu(middle), "start_time_step_index": k}, path)
    init = instantiate_trajectory_initializer(spar, path_to_starting_configuration_data_pickle=str(path))
    assert isinstance(init, StartFromGivenConfigurationTrajectoryInitializer)
    assert init.create_start_time_step_index(T) == k and init.create_end_time_step_index() == 0
    gen2, *_ = _build(name, cases.TRAJECTORIES, cuda, rng_mode="device", seed=21)
    gen2.trajectory_initializer = init
    with torch.no_grad():
        resumed = _np(gen2.sample(B, cuda))                                  # call index 0, as the first run
    assert np.array_equal(resumed.A, full.A)
    assert np.array_equal(resumed.X.view(np.int32), full.X.view(np.int32))
    with pytest.raises(AssertionError):
        gen2.sample(B + 1, cuda)                                             # the file holds B samples (:176-180)


def RS_AXL_cpu(comp):
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    return AXL(A=comp.A.cpu(), X=comp.X.cpu(), L=comp.L.cpu())


@pytest.mark.parametrize("num_atom_types,n_hidden", [(1, 2), (1, 3), (1, 4), (2, 2), (2, 3), (2, 4)])
def test_fused_sampler_register_resident_family(cuda, num_atom_types, n_hidden):
    """The register-resident instantiations of the persistent sampler (N = 8, hidden 64; one or two atom types; 2-4 hidden
    layers; folded input / output layers) against the generic instantiation of the same kernel: the same function, folded
    rounding.  One iteration from the same state: atom types exact, coordinates within 1e-6; a whole trajectory on a neutral
    schedule: atom types exact, coordinates within 1e-5.  And each network really gets its own instantiation."""
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    P = _pkg()
    torch.manual_seed(1234 + n_hidden)
    net = nets.mlp_net(8, num_atom_types, hidden=64, n_hidden=n_hidden).to(cuda)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = P["Noise"](**cases.noise_ns(40, **cases.LIN))
        spar = P["Sampling"](**cases.sampling_ns(8, num_atom_types), rng_mode="device", seed=3, fused_score_network=True)
    gen = P["Langevin"](npar, spar, net)
    with torch.no_grad():
        gen._prepare(cuda)
        gen._begin_call(cuda)
        start = gen.initialize(200, cuda)
        sched, pack = gen._prepare(cuda), gen.fused_pack(cuda)
        assert kernels.lib().mdx_mlp_pc_sample_variant(pack.c_struct, 0) == 100 + 10 * (num_atom_types + 1) + n_hidden
        generic = _hip.MLP_SAMPLE_GENERIC_KERNEL | _hip.MLP_SAMPLE_UNFOLDED
        assert kernels.lib().mdx_mlp_pc_sample_variant(pack.c_struct, generic) == 0
        results = {}
        for name, options in (("family", 0), ("generic", generic)):
            for n_iterations in (1, 40):
                comp = RS.AXL(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
                kernels.mlp_pc_sample(sched, pack, gen._flags(True), 1, False, 40, n_iterations, gen._rng(0), comp.A,
                                      comp.X, comp.L, gen._status, workspace=gen._noise_workspace, options=options)
                results[name, n_iterations] = _np(comp)
    for n_iterations, tol in ((1, 1e-6), (40, 1e-5)):
        a, b = results["family", n_iterations], results["generic", n_iterations]
        assert np.array_equal(a.A, b.A), n_iterations
        assert torus_rel_l2(a.X, b.X) < tol, (n_iterations, torus_rel_l2(a.X, b.X))
    assert (results["family", 40].A != num_atom_types).all()


_SHARD_WORKER = '''
import os, sys, warnings
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
import torch.distributed as dist
import cases, nets
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \\
    PredictorCorrectorSamplingParameters
from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
from diffusion_for_multi_scale_molecular_dynamics_amd.sampling.diffusion_sampling import create_batch_of_samples_sharded
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
device = torch.device("cuda:0")
out = {{}}
for name in {names!r}:
    noise_kw, sampling_kw, netf = cases.TRAJECTORIES[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spar = PredictorCorrectorSamplingParameters(**dict(sampling_kw, number_of_samples={total}, sample_batchsize={batch}),
                                                    rng_mode="device", seed={seed})
    if netf is None:
        net = nets.fake_net(spar.num_atom_types)
    else:
        torch.manual_seed(1234)
        net = netf(None)
    gen = LangevinGenerator(NoiseParameters(**noise_kw), spar, net.to(device))
    with torch.no_grad():
        res = create_batch_of_samples_sharded(gen, spar, device)
    out[name] = {{k: (v if torch.is_tensor(v) else tuple(v)) for k, v in res.items()}}
torch.save({{n: dict(A=o["original_axl"][0].cpu(), X=o["original_axl"][1].cpu(), L=o["original_axl"][2].cpu(),
                    C=o["cartesian_positions"].cpu()) for n, o in out.items()}}, os.path.join({out!r}, f"rank{{rank}}.pt"))
dist.barrier()
dist.destroy_process_group()
print(f"rank {{rank}} ok")
'''


def test_sharded_driver_per_shard_parity_on_the_hip_path(cuda, tmp_path):
    """SURVEY 8(e), 'Seeds / parity under sharding': each rank seeds base + rank, and its shard must equal a single-process
    run with number_of_samples = its share and that seed.  Two fresh child processes (gloo, both on cuda:0) run
    create_batch_of_samples_sharded with the REAL LangevinGenerator (device RNG; the echo network and the small MLP) over
    7 samples in sub-batches of 2 -- rank 0 owns sub-batches 0 and 2 (4 samples), rank 1 owns 1 and 3 (3 samples: uneven);
    the batch both ranks hold after the job's one gather equals, bit for bit, the sub-batches of two single-process
    generators with seeds base + 0 and base + 1, put in sub-batch order (reference loop:
    sampling/diffusion_sampling.py:44-50)."""
    import os
    import subprocess
    import sys
    import warnings
    from conftest import ROOT
    from diffusion_for_multi_scale_molecular_dynamics_amd.sampling.diffusion_sampling import (shard_of_rank, split_sizes)
    names, total, batch, seed = ["traj_fake_c3_m2", "traj_mlp_c3"], 7, 2, 4242
    script = tmp_path / "shard_worker.py"
    script.write_text(_SHARD_WORKER.format(root=ROOT, names=names, total=total, batch=batch, seed=seed, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, f"rank {r} failed:\n{o[-3000:]}"
    got = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    P = _pkg()
    sizes = split_sizes(total, batch)
    assert sizes == [2, 2, 2, 1]
    for name in names:
        noise_kw, sampling_kw, netf = cases.TRAJECTORIES[name]
        pieces = {}
        for r in range(2):                      # the single-process run of rank r's share with seed base + r
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                spar = P["Sampling"](**dict(sampling_kw, number_of_samples=total, sample_batchsize=batch),
                                     rng_mode="device", seed=seed + r)
            if netf is None:
                net = nets.fake_net(spar.num_atom_types)
            else:
                torch.manual_seed(1234)
                net = netf(None)
            gen = P["Langevin"](P["Noise"](**noise_kw), spar, net.to(cuda))
            with torch.no_grad():
                for k, n in shard_of_rank(sizes, r, 2):
                    pieces[k] = gen.sample(n, cuda)
        want_A = torch.cat([pieces[k].A for k in range(len(sizes))]).cpu()
        want_X = torch.cat([pieces[k].X for k in range(len(sizes))]).cpu()
        for r in range(2):                      # every rank holds the whole batch
            g = got[r][name]
            assert torch.equal(g["A"], want_A), (name, r)
            assert torch.equal(g["X"].view(torch.int32), want_X.view(torch.int32)), (name, r)
            assert g["X"].shape[0] == total and torch.equal(g["C"], g["X"] * g["L"][:, None, :3])


# -------------------------------------------------------------------------------------------------------------
# round 5: the sampler around EGNNs with E_GCL's options (attention gate inside the MFMA chain, normalize / tanh in the
# per-node kernel) and around the reference's 1-D template shape
# -------------------------------------------------------------------------------------------------------------
def _option_net(kind, edge_builder=None):
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    common = dict(n_layers=2, coordinate_hidden_dimensions_size=32, coordinate_n_hidden_dimensions=2,
                  message_hidden_dimensions_size=32, message_n_hidden_dimensions=2, node_hidden_dimensions_size=32,
                  node_n_hidden_dimensions=2)
    if kind == "template_1d":        # config_diffusion_egnn_2_atoms_in_1D.yaml:52-67 at a quarter of its width
        p = EGNNScoreNetworkParameters(spatial_dimension=1, num_atom_types=1, normalize=True, edges="fully_connected", **common)
    else:
        p = EGNNScoreNetworkParameters(num_atom_types=2, attention=True, normalize=True, tanh=True, edges="radial_cutoff",
                                       radial_cutoff=7.5, **common)
    return EGNNScoreNetwork(p, edge_builder=edge_builder).eval()


@pytest.mark.parametrize("kind", ["all_options", "template_1d"])
def test_sampler_around_egnn_options_graph_eager_and_oracle(cuda, kind):
    """LangevinGenerator around an EGNN with attention + normalize + tanh (two atom types, radius graph, N = 64) and around the
    reference's 1-D template shape (spatial dimension 1, two atoms, normalize, fully connected): device Philox, five time
    indices, M = 1 -- the iteration replayed from a hipGraph equals the eager launches bit for bit, and both equal the CPU
    oracle's run of the same Philox specification around the same module on the CPU (atom types exact, coordinates <= 1e-5);
    the fused chain ran in every graph layer."""
    P = _pkg()
    import warnings
    if kind == "template_1d":
        skw = cases.sampling_ns(2, 1, M=1, one=False, greedy=False, cell=[1.0], d=1)
        batch = 16
    else:
        skw = cases.sampling_ns(64, 2, M=1, cell=[11.084] * 3)
        batch = 4
    nkw = cases.noise_ns(5, **cases.LIN)
    torch.manual_seed(4321)
    net_cpu = _option_net(kind, edge_builder=nets.oracle_edge_builder)
    outs = {}
    for use_graph in (False, True):
        net = _option_net(kind)
        net.load_state_dict(net_cpu.state_dict())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar, spar = P["Noise"](**nkw), P["Sampling"](**skw, rng_mode="device", seed=99, use_hip_graph=use_graph)
        gen = P["Langevin"](npar, spar, net.to(cuda))
        with torch.no_grad():
            outs[use_graph] = _np(gen.sample(batch, cuda))
        assert gen.f16_range_fallbacks == 0
        assert all(layer._chain[1] is not None for layer in net.egnn.graph_layers), "the fused edge chain did not run"
        if kind == "all_options":
            assert all(layer._chain[1].att_w is not None for layer in net.egnn.graph_layers)
    assert np.array_equal(outs[False].A, outs[True].A)
    assert np.array_equal(outs[False].X.view(np.int32), outs[True].X.view(np.int32))
    ora = RS.OracleLangevinGenerator(npar, spar, net_cpu, noise=RS.PhiloxNoise(99, 0)).sample(batch)
    assert np.array_equal(outs[True].A, ora.A) and (ora.A != spar.num_atom_types).all()
    assert torus_rel_l2(outs[True].X, ora.X) < 1e-5


@pytest.mark.parametrize("seed", range(8 * FUZZ))
def test_sampler_around_random_egnn_configurations_graph_equals_eager(cuda, seed):
    """The sampler around seeded random EGNN configurations (1 - 3 dimensions, 1 - 2 graph layers, widths 16 ... 128 equal or not,
    every option combination, both graph kinds, 1 - 2 atom types, 0 - 2 correctors, random update flags), device Philox, four time
    indices: the iteration replayed from a hipGraph equals the eager launches BIT FOR BIT, the CPU oracle's run of the same
    Philox specification around the same module agrees (atom types exact wherever the logits are not within rounding of a tie:
    checked through the coordinates' <= 1e-4 torus distance and the fraction of equal types), every layer on the fused chain."""
    import warnings
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    P = _pkg()
    rng = np.random.default_rng(7000 + seed)
    d = int(rng.choice([1, 2, 3], p=[0.2, 0.2, 0.6]))
    nat = int(rng.integers(1, 3))
    widths = [int(rng.choice([16, 32, 64, 128]))] * 3 if rng.random() < 0.5 else [int(rng.choice([16, 32, 48, 64])) for _ in range(3)]
    radial = bool(rng.random() < 0.6)
    rc = float(rng.uniform(2.5, 3.5))
    p = EGNNScoreNetworkParameters(
        spatial_dimension=d, num_atom_types=nat, n_layers=int(rng.integers(1, 3)),
        message_hidden_dimensions_size=widths[0], message_n_hidden_dimensions=int(rng.integers(1, 4)),
        coordinate_hidden_dimensions_size=widths[1], coordinate_n_hidden_dimensions=int(rng.integers(1, 4)),
        node_hidden_dimensions_size=widths[2], node_n_hidden_dimensions=int(rng.integers(1, 4)),
        attention=bool(rng.random() < 0.5), tanh=bool(rng.random() < 0.5), normalize=bool(rng.random() < 0.5),
        residual=bool(rng.random() < 0.7), coords_agg=str(rng.choice(["mean", "sum"])), message_agg=str(rng.choice(["mean", "sum"])),
        edges="radial_cutoff" if radial else "fully_connected", radial_cutoff=rc if radial else None)
    N, batch = int(rng.integers(2, 25)), int(rng.integers(1, 5))
    cell = [float(v) for v in rng.uniform(2.2 * rc + 0.5, 2.2 * rc + 4.0, d)]
    skw = cases.sampling_ns(N, nat, M=int(rng.integers(0, 3)), greedy=bool(rng.random() < 0.5), one=bool(rng.random() < 0.5),
                            in_corr=bool(rng.random() < 0.3), cell=cell, d=d)
    nkw = cases.noise_ns(4, **cases.LIN)
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net_cpu = EGNNScoreNetwork(p, edge_builder=nets.oracle_edge_builder if radial else None).eval()
    outs = {}
    for use_graph in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = EGNNScoreNetwork(p).eval()
            net.load_state_dict(net_cpu.state_dict())
            npar, spar = P["Noise"](**nkw), P["Sampling"](**skw, rng_mode="device", seed=5 + seed, use_hip_graph=use_graph)
            gen = P["Langevin"](npar, spar, net.to(cuda))
            with torch.no_grad():
                outs[use_graph] = _np(gen.sample(batch, cuda))
        assert all(layer._chain[1] is not None for layer in net.egnn.graph_layers), ("the fused edge chain did not run", p)
    assert np.array_equal(outs[False].A, outs[True].A), p
    assert np.array_equal(outs[False].X.view(np.int32), outs[True].X.view(np.int32)), p
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ora = RS.OracleLangevinGenerator(npar, spar, net_cpu, noise=RS.PhiloxNoise(5 + seed, 0)).sample(batch)
    assert (ora.A != nat).all() and (outs[True].A != nat).all()
    assert torus_rel_l2(outs[True].X, ora.X) < 1e-4, (torus_rel_l2(outs[True].X, ora.X), p)
    assert (outs[True].A == ora.A).mean() > 0.9, p


def test_use_hip_graph_with_a_network_that_cannot_be_captured_runs_eagerly(cuda):
    """`use_hip_graph: true` around a score network whose forward needs a host read -- an EGNN with a radius graph whose layer
    width (288) is beyond the fused edge chain's widths, so the edge list is sized after reading the edge count -- used to die inside
    the capture (hipErrorStreamCaptureUnsupported).  The network now says so (`capture_safe`), the generator warns once and
    launches the iteration eagerly: the same bits as a generator built with use_hip_graph=False."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    P = _pkg()
    import warnings
    outs = {}
    for use_graph in (True, False):
        torch.manual_seed(77)
        net = EGNNScoreNetwork(EGNNScoreNetworkParameters(
            num_atom_types=1, n_layers=2, coordinate_hidden_dimensions_size=288, coordinate_n_hidden_dimensions=1,
            message_hidden_dimensions_size=288, message_n_hidden_dimensions=1, node_hidden_dimensions_size=32,
            node_n_hidden_dimensions=1, edges="radial_cutoff", radial_cutoff=7.5)).eval().to(cuda)
        assert not net.capture_safe(3, 64, cuda)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            npar = P["Noise"](**cases.noise_ns(4, **cases.LIN))
            spar = P["Sampling"](**cases.sampling_ns(64, 1, M=1, one=False, greedy=False, cell=[10.86] * 3), rng_mode="device",
                                 seed=5, use_hip_graph=use_graph)
        gen = P["Langevin"](npar, spar, net)
        with torch.no_grad():
            if use_graph:
                with pytest.warns(UserWarning, match="launched eagerly"):
                    outs[use_graph] = _np(gen.sample(3, cuda))
                with warnings.catch_warnings():
                    warnings.simplefilter("error")               # (one warning per generator)
                    gen.sample(3, cuda)
            else:
                outs[use_graph] = _np(gen.sample(3, cuda))
        assert "graph_loop" not in gen._buffers
    assert np.array_equal(outs[True].A, outs[False].A)
    assert np.array_equal(outs[True].X.view(np.int32), outs[False].X.view(np.int32))
