"""CPU tests of the host side: ABI surface, loud failure without a GPU, config/factory logic, sharded driver."""
import os
import re
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

import cases
import nets
from conftest import ROOT, load_golden

PKG = "diffusion_for_multi_scale_molecular_dynamics_amd"


def test_library_exports_every_declared_symbol():
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
    _hip.build()
    header = open(os.path.join(ROOT, "include", "mdx_hip.h")).read()
    declared = re.findall(r"MDX_API\s+(?:const\s+char\*|int64_t|int)\s+(mdx_[a-z0-9_]+)\s*\(", header)
    assert len(declared) >= 17 and sorted(declared) == sorted(_hip.ABI_SYMBOLS)
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _hip.LIB_PATH], text=True)
    for sym in declared:
        assert re.search(rf"\sT\s{sym}\b", exported), f"{sym} is declared in include/mdx_hip.h but not exported"
    lib = _hip.lib()                      # loads without a GPU; no compute call is made here
    assert lib.mdx_abi_version() == _hip.ABI_VERSION == 14
    # no vendor GEMM library behind the ABI: every matrix product of the library is a hand-written kernel
    needed = subprocess.check_output(["readelf", "-d", _hip.LIB_PATH], text=True)
    assert "hipblas" not in needed.lower() and "rocblas" not in needed.lower(), needed
    assert lib.mdx_status_string(-2).decode() == "unsupported size or option"
    # the shared object carries gfx950 code
    assert b"gfx950" in open(_hip.LIB_PATH, "rb").read()


def test_no_cpu_fallback():
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
        PredictorCorrectorSamplingParameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
    x = torch.rand(4, 3)
    with pytest.raises(_hip.MdxError, match="no CPU fallback"):
        kernels.relative_coordinates_update(x, x, x, 0.1, 0.1, 0.1)
    gen = LangevinGenerator(NoiseParameters(total_time_steps=3),
                            PredictorCorrectorSamplingParameters(**cases.sampling_ns(4, 1)), nets.fake_net(1))
    with pytest.raises(_hip.MdxError, match="GPU hot path only"):
        gen.sample(2, torch.device("cpu"))
    with pytest.raises(_hip.MdxError):
        kernels.noise_schedule_build(10, "linear", 1e-5, 1e-3, 0.5, 2e-5, 2, "cpu")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no module of the product package may import or load it."""
    pkg_dir = os.path.join(ROOT, PKG)
    pattern = re.compile(r"^\s*(from|import)\s+[\w.]*oracle|mdx_oracle|libmdx_oracle", re.M)
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                assert not pattern.search(open(os.path.join(base, f)).read()), f"{f} refers to the oracle"


def test_parameters_and_factories(tmp_path):
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.constrained_langevin_generator import \
        ConstrainedLangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.instantiate_generator import instantiate_generator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.load_sampling_parameters import \
        load_sampling_parameters
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.sampling_constraint import (
        SamplingConstraint, read_sampling_constraint, write_sampling_constraint)
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.trajectory_initializer import (
        FullRandomTrajectoryInitializer, StartFromGivenConfigurationTrajectoryInitializer,
        instantiate_trajectory_initializer)
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters

    with pytest.raises(AssertionError):
        NoiseParameters(total_time_steps=10, schedule_type="cosine")
    sp = load_sampling_parameters(dict(cases.sampling_ns(8, 1), cell_dimensions=[5.43, 5.43, 5.43]))
    assert sp.number_of_corrector_steps == 1 and sp.small_epsilon == 1e-8 and sp.rng_mode == "reference"
    assert torch.equal(sp.fixed_lattice_parameters, torch.tensor([5.43, 5.43, 5.43, 0, 0, 0]))
    with pytest.raises(AssertionError):
        load_sampling_parameters(dict(cases.sampling_ns(8, 1), algorithm="euler"))
    with pytest.raises(NotImplementedError):
        load_sampling_parameters(dict(cases.sampling_ns(8, 1), algorithm="ode"))
    with pytest.raises(AssertionError):          # fixed lattice without cell dimensions
        load_sampling_parameters(dict(cases.sampling_ns(8, 1), cell_dimensions=None))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        load_sampling_parameters(cases.sampling_ns(8, 1, fixed=False))
        assert any("experimental" in str(x.message) for x in w)

    init = instantiate_trajectory_initializer(sp)
    assert isinstance(init, FullRandomTrajectoryInitializer)
    comp = init.initialize(5, torch.device("cpu"))
    assert (comp.A == 1).all() and comp.X.shape == (5, 8, 3) and ((comp.X >= 0) & (comp.X < 1)).all()
    assert torch.equal(comp.L, sp.fixed_lattice_parameters.repeat(5, 1))
    assert init.create_start_time_step_index(17) == 17 and init.create_end_time_step_index() == 0

    start = dict(noisy_axl=AXL(A=torch.zeros(2, 8, dtype=torch.long), X=torch.rand(2, 8, 3), L=torch.rand(2, 6)),
                 start_time_step_index=4)
    torch.save(start, tmp_path / "start.pkl")
    init2 = instantiate_trajectory_initializer(sp, str(tmp_path / "start.pkl"))
    assert isinstance(init2, StartFromGivenConfigurationTrajectoryInitializer)
    assert init2.create_start_time_step_index(100) == 4
    assert torch.equal(init2.initialize(2, torch.device("cpu")).X, start["noisy_axl"].X)
    with pytest.raises(AssertionError):
        init2.initialize(3, torch.device("cpu"))

    c = SamplingConstraint(elements=["Si"], constrained_relative_coordinates=torch.rand(3, 3),
                           constrained_atom_types=torch.zeros(3, dtype=torch.long))
    write_sampling_constraint(c, tmp_path / "c.pkl")
    c2 = read_sampling_constraint(tmp_path / "c.pkl")
    assert torch.equal(c.constrained_relative_coordinates, c2.constrained_relative_coordinates)
    with pytest.raises(AssertionError):
        SamplingConstraint(elements=["Si"], constrained_relative_coordinates=torch.rand(3, 3),
                           constrained_atom_types=torch.ones(3, dtype=torch.long))
    with pytest.raises(AssertionError):
        SamplingConstraint(elements=["Si"], constrained_relative_coordinates=torch.rand(3, 3).double(),
                           constrained_atom_types=torch.zeros(3, dtype=torch.long))

    npar = NoiseParameters(total_time_steps=5)
    gen = instantiate_generator(sp, npar, nets.fake_net(1), init)
    assert type(gen) is LangevinGenerator
    gen = instantiate_generator(sp, npar, nets.fake_net(1), init, sampling_constraints=c)
    assert type(gen) is ConstrainedLangevinGenerator and torch.equal(gen.constraint_indices, torch.arange(3))
    with pytest.raises(AssertionError):          # more constraints than atoms
        big = SamplingConstraint(elements=["Si"], constrained_relative_coordinates=torch.rand(9, 3),
                                 constrained_atom_types=torch.zeros(9, dtype=torch.long))
        instantiate_generator(sp, npar, nets.fake_net(1), init, sampling_constraints=big)
    with pytest.raises(AssertionError):          # T = 1 is refused like in the reference
        LangevinGenerator(NoiseParameters(total_time_steps=1), sp, nets.fake_net(1))


def test_score_networks_against_reference_forward():
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    for name, net in (("net_mlp_c1", nets.mlp_net(8, 1)), ("net_mlp_c3", nets.mlp_net(8, 2)),
                      ("net_egnn_fc", nets.egnn_net(1, "fully_connected", None)),
                      ("net_egnn_rc", nets.egnn_net(2, "radial_cutoff", 7.5, edge_builder=nets.oracle_edge_builder))):
        g = load_golden(name + ".npz")
        nets.load_fixture_weights(net, g)
        batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.from_numpy(g["A"]), X=torch.from_numpy(g["X"]),
                                            L=torch.from_numpy(g["L"])),
                 TIME: torch.from_numpy(g["time"]), NOISE: torch.from_numpy(g["noise"]),
                 CARTESIAN_FORCES: torch.zeros(g["X"].shape)}
        with torch.no_grad():
            out = net(batch, conditional=False)
        assert torch.isinf(out.A[..., -1]).all() and (out.A[..., -1] < 0).all()       # MASK logit forced to -inf
        np.testing.assert_allclose(out.X.numpy(), g["out_X"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out.A.numpy()[..., :-1], g["out_A"][..., :-1], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out.L.numpy(), g["out_L"], rtol=1e-4, atol=1e-6)


def test_bloch_vectors_and_edges_batch():
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import \
        positive_bloch_wave_vectors
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils.neighbors import get_edges_batch
    assert torch.equal(positive_bloch_wave_vectors(1, 3), torch.eye(3))
    k2 = positive_bloch_wave_vectors(2, 3)
    assert k2.shape == (9, 3) and torch.equal(k2[:3], torch.eye(3)) and ((k2[3:] ** 2).sum(1) == 2).all()
    e = get_edges_batch(4, 3)
    assert e.shape == (3 * 4 * 3, 2) and (e[:, 0] != e[:, 1]).all() and (e[:, 0] // 4 == e[:, 1] // 4).all()
    key = e[:, 0] * 100 + e[:, 1]
    assert (key[1:] > key[:-1]).all()


def test_sample_trajectory_round_trip(tmp_path):
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils.sample_trajectory import SampleTrajectory
    rec = SampleTrajectory()
    rec.record("noise_parameters", dict(total_time_steps=3))
    rec.record("predictor_step", dict(time_step_index=3))
    rec.record("predictor_step", dict(time_step_index=2))
    rec.write_to_pickle(tmp_path / "t.pt")
    data = torch.load(tmp_path / "t.pt", weights_only=False)
    assert data["noise_parameters"] == dict(total_time_steps=3)          # single entries are unwrapped
    assert [e["time_step_index"] for e in data["predictor_step"]] == [3, 2]


def test_split_and_shard_bookkeeping():
    from diffusion_for_multi_scale_molecular_dynamics_amd.sampling.diffusion_sampling import shard_of_rank, split_sizes
    assert split_sizes(16, None) == [16] and split_sizes(16, 2) == [2] * 8 and split_sizes(7, 3) == [3, 3, 1]
    sizes = split_sizes(23, 4)
    got = sorted(sum((shard_of_rank(sizes, r, 4) for r in range(4)), []))
    assert got == list(enumerate(sizes))


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from types import SimpleNamespace
from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
from diffusion_for_multi_scale_molecular_dynamics_amd.sampling.diffusion_sampling import (
    create_batch_of_samples, create_batch_of_samples_sharded)

class DummyGenerator:
    # like the reference's tests/sampling/test_diffusion_sampling.py DummyGenerator: deterministic per call index
    def __init__(self): self.calls = []
    def sample(self, n, device):
        k = len(self.calls); self.calls.append(n)
        g = torch.Generator().manual_seed(1000 + n)      # content depends on the sub-batch size only
        return AXL(A=torch.randint(0, 2, (n, 4), generator=g), X=torch.rand(n, 4, 3, generator=g),
                   L=torch.rand(n, 6, generator=g))

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
sp = SimpleNamespace(number_of_samples=11, sample_batchsize=2, number_of_atoms=4, spatial_dimension=3)
gen = DummyGenerator()
out = create_batch_of_samples_sharded(gen, sp, torch.device("cpu"))
single = create_batch_of_samples(DummyGenerator(), sp, torch.device("cpu"))
assert out["original_axl"].A.shape == (11, 4) and out["cartesian_positions"].shape == (11, 4, 3)
assert torch.equal(out["original_axl"].A, single["original_axl"].A)
assert torch.equal(out["original_axl"].X, single["original_axl"].X)
assert torch.equal(out["cartesian_positions"], single["cartesian_positions"])
assert (out["original_axl"].L[:, 3:] == 0).all()
assert sum(gen.calls) == sum(n for k, n in enumerate([2, 2, 2, 2, 2, 1]) if k % world == rank)
# batch statistics of the adaptive corrector across shards == the un-sharded means (one 4-scalar all-reduce)
from diffusion_for_multi_scale_molecular_dynamics_amd.utils.batch_statistics import global_means
g = torch.Generator().manual_seed(7)
full_a, full_b = torch.rand(10, generator=g), torch.rand(10, 8, generator=g)
lo, hi = (0, 3) if rank == 0 else (3, 10)          # ragged shards
ma, mb = global_means(full_a[lo:hi], full_b[lo:hi], across_ranks=True)
assert torch.allclose(ma, full_a.mean(), rtol=1e-6) and torch.allclose(mb, full_b.mean(), rtol=1e-6)
la, lb = global_means(full_a[lo:hi], full_b[lo:hi], across_ranks=False)
assert torch.equal(la, full_a[lo:hi].mean()) and torch.equal(lb, full_b[lo:hi].mean())
# trajectories.pt of a sharded run == the file a single process writes (every rank's sub-batches, in sub-batch order)
from pathlib import Path
from diffusion_for_multi_scale_molecular_dynamics_amd.sample_diffusion import write_trajectories
from diffusion_for_multi_scale_molecular_dynamics_amd.utils.sample_trajectory import SampleTrajectory

class RecordingGenerator(DummyGenerator):
    def __init__(self):
        super().__init__()
        self.sample_trajectory_recorder = SampleTrajectory()
        self.sample_trajectory_recorder.record(key="noise_parameters", entry=dict(total_time_steps=3))
    def sample(self, n, device):
        out = super().sample(n, device)
        for i in (3, 2, 1):
            self.sample_trajectory_recorder.record(key="predictor_step", entry=dict(time_step_index=i, composition_i=out))
            for m in range(2):
                self.sample_trajectory_recorder.record(key="corrector_step", entry=dict(time_step_index=i - 1, composition_i=out))
        return out

sp = SimpleNamespace(number_of_samples=9, sample_batchsize=2, number_of_atoms=4, spatial_dimension=3, record_samples=True)
out_dir = Path({out!r})
gen = RecordingGenerator()
create_batch_of_samples_sharded(gen, sp, torch.device("cpu"))
write_trajectories(gen.sample_trajectory_recorder, sp, out_dir)
if rank == 0:
    single = RecordingGenerator()
    create_batch_of_samples(single, sp, torch.device("cpu"))
    single.sample_trajectory_recorder.write_to_pickle(out_dir / "single.pt")
    got = torch.load(out_dir / "trajectories.pt", weights_only=False)
    want = torch.load(out_dir / "single.pt", weights_only=False)
    assert sorted(got) == sorted(want) and got["noise_parameters"] == want["noise_parameters"]
    for key in ("predictor_step", "corrector_step"):
        assert len(got[key]) == len(want[key]) == 5 * (3 if key == "predictor_step" else 6)
        for a, b in zip(got[key], want[key]):
            assert a["time_step_index"] == b["time_step_index"]
            assert all(torch.equal(x, y) for x, y in zip(a["composition_i"], b["composition_i"]))
    assert not list(out_dir.glob("trajectories.rank*.pt"))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_driver_gloo_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_pack_and_unpack_compositions_round_trip():
    """A, X, L of a structure as one byte row: what the job's single all-gather moves."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    from diffusion_for_multi_scale_molecular_dynamics_amd.sampling.diffusion_sampling import (pack_compositions,
                                                                                              unpack_compositions)
    g = torch.Generator().manual_seed(3)
    for batch, n, d in ((5, 8, 3), (0, 8, 3), (2, 1, 2), (3, 216, 3)):
        comp = AXL(A=torch.randint(0, 3, (batch, n), generator=g), X=torch.rand(batch, n, d, generator=g),
                   L=torch.rand(batch, d * (d + 1) // 2, generator=g))
        rows = pack_compositions(comp)
        assert rows.dtype == torch.uint8 and rows.shape == (batch, 8 * n + 4 * n * d + 4 * (d * (d + 1) // 2))
        back = unpack_compositions(rows, n, d)
        assert all(torch.equal(a, b) for a, b in zip(comp, back))
        two = unpack_compositions(torch.stack([rows, rows]), n, d)              # leading dimensions are kept
        assert two.X.shape == (2, batch, n, d) and torch.equal(two.A[1], comp.A)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns torch.distributed.run as a child (before any GPU
    initialisation), relays rank 0's JSON line and exits with the child's code.  Rehearsal mode: host tensors, gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    import json
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["gather_ok"] and line["collectives"] == 1
    # the per-rank record (one small all-gather after the timed region): every rank's own figures, the slowest rank named,
    # and what one card did for comparison with an N = 1 line
    assert [r["rank"] for r in line["per_rank"]] == [0, 1]
    assert all(set(r) == {"rank", "ms_per_step", "trajectory_ms", "sclk_mhz_mean", "power_w_mean", "power_cap_w",
                          "sensor_samples", "f16_range_fallbacks"} for r in line["per_rank"])
    assert [r["ms_per_step"] for r in line["per_rank"]] == [1.0, 2.0] and line["slowest_rank"] == 1
    assert line["n1_equivalent"]["fastest_rank"] == 0 and line["n1_equivalent"]["spread"] > 0
    assert line["n1_equivalent"]["value_per_gpu_fastest_rank"] > line["n1_equivalent"]["value_per_gpu_slowest_rank"] > 0
    # a failing child is reported through the exit code
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch",
                          "--workload", "nope"], env=env, capture_output=True, text=True, timeout=240)
    assert bad.returncode != 0


@pytest.mark.parametrize("how", ["raise", "kill"])
def test_bench_fails_fast_when_a_rank_dies(how):
    """One of two ranks raises -- or is killed by a signal -- right after the rendezvous, while the other is on its way into the
    job's collective: the launcher exits non-zero within seconds (torch.distributed.run ends the surviving rank; every
    collective carries a 120 s time-out besides), names the failed rank with its message on stderr, and prints no JSON line.
    No rendezvous port is given: a free one is picked."""
    import json
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch",
                          "--rehearse-fail", f"{how}:1"], env=env, capture_output=True, text=True, timeout=200)
    elapsed = time.perf_counter() - t0
    assert out.returncode != 0 and elapsed < 150, (out.returncode, elapsed)
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout
    report = [json.loads(ln) for ln in out.stderr.splitlines() if ln.startswith('{"bench_launcher"')]
    assert len(report) == 1 and report[0]["bench_launcher"]["exit_code"] == out.returncode
    errors = report[0]["bench_launcher"]["rank_errors"]
    if how == "raise":
        # (rank 0 may report too -- its collective fails when its peer is gone -- but the rank that failed first is listed first)
        assert errors[0]["rank"] == 1 and "rank 1 was asked to fail" in errors[0]["error"]
    else:
        assert errors == ["no rank left a report (killed by a signal?)"]


def test_cif_and_xyz_writers(tmp_path):
    """samples.pt / recorded trajectories -> CIF and extended-XYZ files with the reference's naming and XYZ header
    (analysis/ovito_utilities/trajectory_io.py:24-140): the files parse back to the same structures."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.analysis.ovito_utilities import trajectory_io as io
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    g = torch.Generator().manual_seed(2)
    samples, steps, n = 3, 4, 5
    traj = AXL(A=torch.randint(0, 3, (samples, steps, n), generator=g), X=torch.rand(samples, steps, n, 3, generator=g),
               L=torch.tensor([5.0, 6.0, 7.0, 0, 0, 0]).repeat(samples, steps, 1))
    props = {"uncertainty": torch.rand(samples, steps, n, 1, generator=g)}
    io.create_cif_files(["Si", "Ge"], tmp_path, 1, traj)
    io.create_xyz_files(["Si", "Ge"], tmp_path, 1, traj, props)
    cif_dir, xyz_dir = tmp_path / "cif_files_trajectory_1", tmp_path / "xyz_files_trajectory_1"
    assert sorted(p.name for p in cif_dir.iterdir()) == [f"diffusion_positions_step_{k}.cif" for k in range(steps)]
    assert sorted(p.name for p in xyz_dir.iterdir()) == [f"diffusion_positions_step_{k}.xyz" for k in range(steps)]
    symbols = {0: "Ge", 1: "Si", 2: "X"}                       # sorted element names, MASK -> X
    text = (cif_dir / "diffusion_positions_step_2.cif").read_text()
    assert "_cell_length_a   5.00000000" in text and "_cell_angle_gamma   90.00000000" in text
    rows = [ln.split() for ln in text.splitlines() if ln.startswith("  ") and len(ln.split()) == 7]
    assert [r[0] for r in rows] == [symbols[int(a)] for a in traj.A[1, 2]]
    got = torch.tensor([[float(v) for v in r[3:6]] for r in rows])
    assert torch.allclose(got, traj.X[1, 2], atol=1e-7)
    lines = (xyz_dir / "diffusion_positions_step_3.xyz").read_text().splitlines()
    assert lines[0] == str(n)
    assert lines[1] == 'Lattice="5.0 0.0 0.0 0.0 6.0 0.0 0.0 0.0 7.0" Origin="0 0 0" pbc="T T T" ' \
                       'Properties=pos:R:3:uncertainty:R:1'
    body = torch.tensor([[float(v) for v in ln.split()] for ln in lines[2:]])
    assert torch.allclose(body[:, :3], (traj.X[1, 3] * torch.tensor([5.0, 6.0, 7.0])).double().float(), atol=1e-6)
    assert torch.allclose(body[:, 3], props["uncertainty"][1, 3, :, 0], atol=1e-7)
    # samples.pt of sample_diffusion -> one file per structure
    final = AXL(A=traj.A[:, -1], X=traj.X[:, -1], L=traj.L[:, -1])
    torch.save({"cartesian_positions": final.X * 5.0, "original_axl": final}, tmp_path / "samples.pt")
    io.write_samples(tmp_path / "samples.pt", ["Si", "Ge"], tmp_path / "out", format="xyz")
    assert len(list((tmp_path / "out" / "xyz_files_trajectory_0").iterdir())) == samples
    with pytest.raises(NotImplementedError):
        io.create_io_files(["Si"], tmp_path, None, final, None, "pdb")


def test_edge_chain_instantiations_keep_their_request_form_valid():
    """The production-size piece-sums instantiations of the edge chain issue their weight-stream requests without the guard
    wait states that protect a scalar register restored from a spill (csrc/mdx_egnn_chain.hip, issue_piece): they must not
    spill scalar registers.  The BUILD enforces it (csrc/Makefile runs check_chain_resources.py on the compiler's resource
    remarks of that very compilation and deletes the object otherwise); here: the remarks of the library under test pass the
    check, and the check really fails on a spill."""
    import subprocess
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
    _hip.build()
    csrc = os.path.dirname(_hip.LIB_PATH)
    remarks, checker = os.path.join(csrc, "mdx_egnn_chain.remarks.txt"), os.path.join(csrc, "check_chain_resources.py")
    if not os.path.exists(remarks):
        pytest.skip("the library was built elsewhere (no compiler remarks next to it)")
    assert os.path.getmtime(remarks) >= os.path.getmtime(os.path.join(csrc, "mdx_egnn_chain.hip")) - 1, "stale remarks: rebuild"
    out = subprocess.run([sys.executable, checker, remarks], capture_output=True, text=True)
    assert out.returncode == 0 and "no scalar-register spills" in out.stdout, out.stdout + out.stderr
    text = open(remarks).read()
    assert "egnn_edge_chain_kernelILi256ELi2ELi2E" in text and "egnn_edge_chain_kernelILi256ELi1ELi2E" in text
    # a doctored copy with a spill in a <256, PREC, 2> kernel must be refused
    import re
    import tempfile
    k = text.index("egnn_edge_chain_kernelILi256ELi2ELi2ELb0E")
    doctored = text[:k] + re.sub(r"SGPRs Spill: 0", "SGPRs Spill: 3", text[k:], count=1)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(doctored)
    bad = subprocess.run([sys.executable, checker, f.name], capture_output=True, text=True)
    os.unlink(f.name)
    assert bad.returncode != 0 and "<256,2,2>" in (bad.stdout + bad.stderr)


VARIANTS = {"attention": dict(attention=True), "normalize_tanh": dict(normalize=True, tanh=True),
            "sum_noresidual": dict(coords_agg="sum", message_agg="sum", residual=False),
            "all_duplicates_kept": dict(attention=True, normalize=True, tanh=True, drop_duplicate_edges=False)}


def variant_tolerance(g, name):
    """1e-5 (north_star) -- or, where the REFERENCE's own fp32 output is further than that from the exact (fp64) evaluation of
    the same network on the same inputs, that distance: a small normalised network with attention and tanh is badly
    conditioned (the reference sits 7e-5 from fp64 on `all_duplicates_kept`, 1.2e-5 on `normalize_tanh`), and no fp32
    evaluation in another order can be asked to land closer to the reference than the exact answer does."""
    from oracle import mdx_oracle
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    mdx_oracle.build()
    net, batch = variant_case(g, name, variant_net(name, edge_builder=nets.oracle_edge_builder))
    net = net.double()
    batch = {k: (AXL(A=v.A, X=v.X.double(), L=v.L.double()) if isinstance(v, tuple) else v.double()) for k, v in batch.items()}
    with torch.no_grad():
        exact = net(batch, conditional=False).X.numpy()
    ref = g[f"{name}/out_X"].astype(np.float64)
    return max(1e-5, float(np.linalg.norm(ref - exact) / np.linalg.norm(ref)))


def variant_net(name, edge_builder=None):
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    p = EGNNScoreNetworkParameters(num_atom_types=2, n_layers=2, coordinate_hidden_dimensions_size=32,
                                   coordinate_n_hidden_dimensions=2, message_hidden_dimensions_size=32,
                                   message_n_hidden_dimensions=2, node_hidden_dimensions_size=32, node_n_hidden_dimensions=2,
                                   edges="radial_cutoff", radial_cutoff=7.5, **VARIANTS[name])
    return EGNNScoreNetwork(p, edge_builder=edge_builder).eval()


def variant_case(g, name, net, device="cpu"):
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    prefix = name + "/net/"
    net.load_state_dict({k[len(prefix):]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith(prefix)})
    t = lambda key: torch.from_numpy(g[f"{name}/{key}"]).to(device)        # noqa: E731
    batch = {NOISY_AXL_COMPOSITION: AXL(A=t("A"), X=t("X"), L=t("L")), TIME: t("time"), NOISE: t("noise"),
             CARTESIAN_FORCES: torch.zeros(g[f"{name}/X"].shape, device=device)}
    return net.to(device), batch


@pytest.mark.parametrize("name", list(VARIANTS))
def test_egnn_option_variants_against_reference_forward(name):
    """E_GCL's options beyond the BASELINE configurations -- attention, normalize, tanh, sum aggregations, no residual,
    drop_duplicate_edges=False (models/egnn.py:36-66,128-131,157,234-264; models/egnn_utils.py:111-140) -- the product's module on
    the CPU (oracle edge list) against the REFERENCE's forward on the same weights (tests/golden/net_egnn_variants.npz)."""
    from oracle import mdx_oracle
    mdx_oracle.build()
    g = load_golden("net_egnn_variants.npz")
    net, batch = variant_case(g, name, variant_net(name, edge_builder=nets.oracle_edge_builder))
    with torch.no_grad():
        out = net(batch, conditional=False)
    ref = g[f"{name}/out_X"].astype(np.float64)
    # (with drop_duplicate_edges=False the reference sums a node's edges in ITS list order -- by periodic image -- and the product
    # in sorted order: the same multiset, test_clipped_cell_has_no_duplicate_edges, another fp32 summation order)
    assert np.linalg.norm(out.X.numpy() - ref) / np.linalg.norm(ref) < variant_tolerance(g, name)
    np.testing.assert_allclose(out.A.numpy()[..., :-1], g[f"{name}/out_A"][..., :-1], rtol=1e-4, atol=1e-5)


WIDE_OPTIONS = {
    # name: (hyper-parameters, formula scale) -- tests/golden/make_golden.py::golden_egnn_options_wide
    "template_1d": (dict(spatial_dimension=1, num_atom_types=1, n_layers=4, coordinate_hidden_dimensions_size=128,
                         coordinate_n_hidden_dimensions=4, coords_agg="mean", message_hidden_dimensions_size=128,
                         message_n_hidden_dimensions=4, node_hidden_dimensions_size=128, node_n_hidden_dimensions=4,
                         attention=False, normalize=True, residual=True, tanh=False, edges="fully_connected"), 2.0),
    "attention_256": (dict(num_atom_types=2, n_layers=2, coordinate_hidden_dimensions_size=256, coordinate_n_hidden_dimensions=2,
                           message_hidden_dimensions_size=256, message_n_hidden_dimensions=2, node_hidden_dimensions_size=256,
                           node_n_hidden_dimensions=2, attention=True, tanh=True, edges="radial_cutoff", radial_cutoff=7.5), 2.0),
    "normalize_128": (dict(num_atom_types=2, n_layers=2, coordinate_hidden_dimensions_size=128, coordinate_n_hidden_dimensions=3,
                           message_hidden_dimensions_size=128, message_n_hidden_dimensions=3, node_hidden_dimensions_size=128,
                           node_n_hidden_dimensions=3, attention=True, normalize=True, coords_agg="sum", message_agg="sum",
                           edges="radial_cutoff", radial_cutoff=7.5), 1.5),
    # narrow / unequal widths: the reference's default hyper-parameters (message 16, node 32, coordinate 32); 48 / 64 / 96
    "default_widths": (dict(num_atom_types=2, edges="radial_cutoff", radial_cutoff=7.5), 2.0),
    "unequal_48_96": (dict(num_atom_types=2, n_layers=2, coordinate_hidden_dimensions_size=96, coordinate_n_hidden_dimensions=2,
                           message_hidden_dimensions_size=48, message_n_hidden_dimensions=2, node_hidden_dimensions_size=64,
                           node_n_hidden_dimensions=2, attention=True, tanh=True, edges="radial_cutoff", radial_cutoff=7.5), 2.0),
}


def wide_option_case(g, name, device="cpu", edge_builder=None):
    """(network with the formula weights of the fixture, batch) of one case of tests/golden/net_egnn_options_wide.npz"""
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import (AXL, CARTESIAN_FORCES, NOISE,
                                                                              NOISY_AXL_COMPOSITION, TIME)
    from formula_weights import fill_with_formula
    kw, scale = WIDE_OPTIONS[name]
    assert float(g[f"{name}/formula_scale"]) == scale
    net = fill_with_formula(EGNNScoreNetwork(EGNNScoreNetworkParameters(**kw), edge_builder=edge_builder).eval(), scale=scale)
    t = lambda key: torch.from_numpy(g[f"{name}/{key}"]).to(device)        # noqa: E731
    batch = {NOISY_AXL_COMPOSITION: AXL(A=t("A"), X=t("X"), L=t("L")), TIME: t("time"), NOISE: t("noise"),
             CARTESIAN_FORCES: torch.zeros(g[f"{name}/X"].shape, device=device)}
    return net.to(device), batch


@pytest.mark.parametrize("name", list(WIDE_OPTIONS))
def test_egnn_options_at_kernel_widths_against_reference_forward(name):
    """E_GCL's options at widths 128 / 256 -- the reference's shipped 1-D template (normalize=True, hidden 128, d = 1:
    configuration_templates/.../config_diffusion_egnn_2_atoms_in_1D.yaml:52-67), attention + tanh at 256, attention + normalize
    with sum aggregations at 128 -- the product's module on the CPU against the REFERENCE's forward on the same formula
    weights: scores <= 1e-5 rel-L2, logits close."""
    from oracle import mdx_oracle
    mdx_oracle.build()
    g = load_golden("net_egnn_options_wide.npz")
    net, batch = wide_option_case(g, name, edge_builder=nets.oracle_edge_builder)
    with torch.no_grad():
        out = net(batch, conditional=False)
    ref = g[f"{name}/out_X"].astype(np.float64)
    assert np.linalg.norm(out.X.numpy() - ref) / np.linalg.norm(ref) < 1e-5
    np.testing.assert_allclose(out.A.numpy()[..., :-1], g[f"{name}/out_A"][..., :-1], rtol=1e-4, atol=1e-5)


def test_d3pm_utils_against_reference_golden():
    """utils/d3pm_utils.py (the callables a reference-style plugin imports, src/.../utils/d3pm_utils.py:7-150) on host tensors
    against what the reference computed on the same operands (tests/golden/make_golden.py::golden_d3pm_utils)."""
    from conftest import load_golden
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils import d3pm_utils as D
    g = load_golden("d3pm_utils.npz")
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k]))          # noqa: E731
    B, N, C = g["onehot"].shape
    onehot = D.class_index_to_onehot(t("index"), C)
    assert onehot.dtype == torch.float32 and torch.equal(onehot, t("onehot"))
    shared = [t(k).expand(B, N, C, C) for k in ("q", "q_bar", "q_bar_tm1")]
    atoms = [t(k) for k in ("q_atoms", "q_bar_atoms", "q_bar_tm1_atoms")]
    close = lambda a, k: np.testing.assert_allclose(a.numpy(), g[k], rtol=2e-6, atol=1e-9)      # noqa: E731
    close(D.compute_q_at_given_a0(onehot, atoms[1]), "q_at_given_a0")
    close(D.compute_q_at_given_a0(t("soft"), shared[1]), "q_at_given_a0_soft")
    close(D.compute_q_at_given_atm1(onehot, atoms[0]), "q_at_given_atm1")
    close(D.get_probability_from_logits(t("logits"), 1e-8), "probability_from_logits")
    close(D.get_probability_at_previous_time_step(t("logits"), onehot, *shared, small_epsilon=1e-8,
                                                  probability_at_zeroth_timestep_are_logits=True), "previous_logits_shared")
    close(D.get_probability_at_previous_time_step(t("logits"), onehot, *atoms, small_epsilon=1e-8,
                                                  probability_at_zeroth_timestep_are_logits=True), "previous_logits_atoms")
    close(D.get_probability_at_previous_time_step(t("soft"), onehot, *shared, small_epsilon=1e-8), "previous_soft_shared")


@pytest.mark.parametrize("options", [dict(), dict(attention=True, tanh=True), dict(normalize=True, coords_agg="sum", message_agg="sum", residual=False)])
def test_e_gcl_public_pieces_compose_to_forward(options):
    """E_GCL's public sub-methods carry the reference's names and signatures (src/models/egnn.py:136-262: message_model,
    node_model, coord_model -- in place --, coord2radial, normalize_radial_norm); composed the way the reference's forward
    composes them (:264-289) on an UNSORTED edge list they give what this package's forward gives on the sorted one."""
    import torch
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.egnn import E_GCL
    torch.manual_seed(3)
    layer = E_GCL(input_size=16, message_n_hidden_dimensions=2, message_hidden_dimensions_size=24, node_n_hidden_dimensions=2,
                  node_hidden_dimensions_size=20, coordinate_n_hidden_dimensions=2, coordinate_hidden_dimensions_size=28,
                  output_size=16, **options).double().eval()
    n = 9
    h, coord = torch.randn(n, 16, dtype=torch.float64), torch.randn(n, 4, dtype=torch.float64)
    edges = torch.tensor([(i, j) for i in range(n) for j in range(n) if i != j and (i + 2 * j) % 3])
    shuffled = edges[torch.randperm(edges.shape[0])]
    with torch.no_grad():
        want_h, want_x = layer(h, edges, coord.clone())
        radial, coord_diff = layer.coord2radial(shuffled, coord)
        messages = layer.message_model(h[shuffled[:, 0]], h[shuffled[:, 1]], radial)
        moved = coord.clone()
        returned = layer.coord_model(moved, shuffled, coord_diff, messages)
        got_h = layer.node_model(h, shuffled, messages)
    assert returned is moved                                            # in place, like the reference
    assert torch.allclose(got_h, want_h, rtol=1e-12, atol=1e-12) and torch.allclose(moved, want_x, rtol=1e-12, atol=1e-12)
    r2 = torch.tensor([[0.0], [1e-3], [4.0]], dtype=torch.float64)
    assert torch.allclose(layer.normalize_radial_norm(r2), torch.tanh(r2) / torch.sqrt(r2 + layer.epsilon ** 2))


def test_small_modules_under_the_reference_paths():
    """Helpers the reference's callers import by module path: sigma calculators (noise_schedulers/sigma_calculator.py) against
    the closed forms and their derivatives, the 27 image vectors in itertools.product order (utils/lattice_utils.py:10-29), the
    unsorted segment reductions and the fully connected edge list (models/egnn_utils.py:11-82), ElementTypes
    (data/element_types.py), the geometry helpers of utils/basis_transformations.py."""
    import itertools
    import numpy as np
    import torch
    from diffusion_for_multi_scale_molecular_dynamics_amd.data.element_types import NULL_ELEMENT, NULL_ELEMENT_ID, ElementTypes
    from diffusion_for_multi_scale_molecular_dynamics_amd.models import egnn_utils, graph_utils
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.sigma_calculator import (
        ExponentialSigmaCalculator, LinearSigmaCalculator, instantiate_sigma_calculator)
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils import basis_transformations as bt, lattice_utils, neighbors
    t = torch.linspace(0, 1, 7)
    exp = instantiate_sigma_calculator(1e-3, 0.5, "exponential")
    lin = instantiate_sigma_calculator(1e-3, 0.5, "linear")
    assert isinstance(exp, ExponentialSigmaCalculator) and isinstance(lin, LinearSigmaCalculator)
    assert torch.allclose(exp(t), 1e-3 * (0.5 / 1e-3) ** t) and torch.allclose(lin(t), 1e-3 + (0.5 - 1e-3) * t)
    assert torch.allclose(exp.get_sigma_time_derivative(t), np.log(500.0) * exp(t)) and torch.allclose(lin.get_sigma_time_derivative(t), torch.full_like(t, 0.499))
    assert sorted(exp.state_dict()) == ["log_ratio", "ratio", "sigma_max", "sigma_min"] and sorted(lin.state_dict()) == ["sigma_difference", "sigma_max", "sigma_min"]
    with pytest.raises(NotImplementedError, match="not implemented"):
        instantiate_sigma_calculator(1e-3, 0.5, "cosine")
    vectors = lattice_utils.get_relative_coordinates_lattice_vectors(1, 3)
    assert vectors.dtype == torch.float32 and vectors.tolist() == [list(map(float, v)) for v in itertools.product((-1, 0, 1), repeat=3)]
    assert lattice_utils.get_relative_coordinates_lattice_vectors(2, 2).shape == (25, 2)
    bloch = lattice_utils.get_cubic_point_group_positive_normalized_bloch_wave_vectors(1, 3)
    assert bloch.dtype == torch.int32 and bloch.tolist() == [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
    data, ids = torch.arange(12.0).reshape(6, 2), torch.tensor([2, 0, 2, 2, 0, 3])
    assert egnn_utils.unsorted_segment_sum(data, ids, 5).tolist() == [[10.0, 12.0], [0.0, 0.0], [10.0, 13.0], [10.0, 11.0], [0.0, 0.0]]
    assert torch.allclose(egnn_utils.unsorted_segment_mean(data, ids, 5),
                          torch.tensor([[5.0, 6.0], [0.0, 0.0], [10 / 3, 13 / 3], [10.0, 11.0], [0.0, 0.0]]))
    assert egnn_utils.get_edges(3) == [[0, 1], [0, 2], [1, 0], [1, 2], [2, 0], [2, 1]]
    assert egnn_utils.get_edges_batch is neighbors.get_edges_batch and graph_utils.get_adj_matrix is neighbors.get_adj_matrix
    elements = ElementTypes(["Si", "Ge"])
    assert elements.elements == ["Ge", "Si"] and elements.element_ids == [0, 1] and elements.number_of_atom_types == 2
    assert elements.get_element_id("Si") == 1 and elements.get_element(0) == "Ge"
    assert elements.get_element(NULL_ELEMENT_ID) == NULL_ELEMENT and elements.get_element_id(NULL_ELEMENT) == NULL_ELEMENT_ID == -1
    cell = torch.tensor([[[4.0, 0.0, 0.0], [0.5, 5.0, 0.0], [0.0, 0.2, 6.0]]])
    x = torch.rand(1, 7, 3)
    assert torch.allclose(bt.get_relative_coordinates_from_cartesian_positions(bt.get_positions_from_coordinates(x, cell),
                                                                                bt.get_reciprocal_basis_vectors(cell)), x, atol=1e-6)
    assert [bt.get_spatial_dimension_from_number_of_lattice_parameters(k) for k in (1, 3, 6)] == [1, 2, 3]
    assert bt.map_unit_cell_to_lattice_parameters(np.diag([1.0, 2.0, 3.0]), engine="numpy").tolist() == [1, 2, 3, 0, 0, 0]
    assert bt.map_numpy_unit_cell_to_lattice_parameters(np.diag([1.0, 2.0])).tolist() == [1, 2, 0]
    noisy = torch.tensor([[2.0, 5.0, 7.0, 0.3, -0.2, 9.0]])
    assert bt.map_noisy_axl_lattice_parameters_to_unit_cell_vectors(noisy).tolist() == [[[4.0, 0, 0], [0, 5.0, 0], [0, 0, 7.0]]]
    assert noisy[0, 0] == 2.0                                                                   # the caller's tensor is not touched


def test_host_draws_do_not_depend_on_the_thread_count():
    """The parity mode draws on ONE host thread (generators/noise_sources.one_host_thread: the machine-sized intra-op pool costs
    7 ms per C3 iteration): torch's CPU generator and the Gumbel transform give the same bits with 1, 2 and 8 threads, the
    caller's thread setting is restored, and ReferenceOrderNoise consumes the global stream exactly as bare torch calls do."""
    import torch
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.noise_sources import ReferenceOrderNoise, one_host_thread
    before = torch.get_num_threads()
    outs = []
    try:
        for n in (1, 2, 8):
            torch.set_num_threads(n)
            torch.manual_seed(5)
            a, b, c = torch.randn(512, 64, 3), torch.rand(512, 64, 2), torch.randn(512, 6)
            outs.append((a, b, c, -torch.log(-torch.log(b.clip(min=1e-8)))))
        assert all(all(torch.equal(x, y) for x, y in zip(outs[0], other)) for other in outs[1:])
        torch.set_num_threads(4)
        with one_host_thread():
            assert torch.get_num_threads() == 1
        assert torch.get_num_threads() == 4
        torch.manual_seed(5)
        source = ReferenceOrderNoise()
        drawn = (source.randn(512, 64, 3), source.rand(512, 64, 2), source.randn(512, 6))
        assert all(torch.equal(x, y) for x, y in zip(drawn, outs[0][:3])) and torch.get_num_threads() == 4
    finally:
        torch.set_num_threads(before)
